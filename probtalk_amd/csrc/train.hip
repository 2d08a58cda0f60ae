// Training-mode kernels of the predictor path (fp32): weight/bias gradients of the tap-GEMM, LayerNorm and the
// stylization modulate+SiLU forward/backward, attention backward, cross-entropy, train-mode BatchNorm statistics and
// application, gate / activation derivatives, embedding gradient, dropout, gradient norm and AdamW.
// The forward GEMMs and the input-gradient GEMMs are the inference tap-GEMM (gemm.hip) fed with re-laid-out weights:
//   dX of Linear / stride-1 conv = tap-GEMM of dY with W^T per tap and negated tap offsets,
//   dX of the k4 s2 conv         = the ConvTranspose form (two output parities x two taps) with W^T per tap,
//   dX of the k4 s2 transposed conv = the k4 s2 conv form.
// Reference lines: predictor.py:168-280 (train step), PTransformer.py:443-504 (train forward), 28-165 (style block),
// vqvae_modules.py:107-200 (ConvNormRelu with batch statistics), torch.optim.AdamW, torch.nn.utils.clip_grad_norm_.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "pt_common.h"

namespace pt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DM = 512;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ unsigned mix32(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (unsigned)((z ^ (z >> 31)) >> 32);
}
// dropout keep-factor of attention weight (b,h,i,j): 0 or 1/(1-p), from a counter-based hash (replayed in backward)
__device__ __forceinline__ float attn_keep(unsigned long long seed, size_t idx, unsigned thr, float keep_scale) {
  return mix32(seed * 0x100000001B3ull + idx) >= thr ? keep_scale : 0.f;
}

// ---------------------------------------------------------------------------
// Weight gradient of the tap-GEMM: dW[t][n][c] += sum_m dY[orow(m)][n] * X[src(m,t)][c]
// (v_mfma_f32_32x32x2_f32 with the GEMM row index m as the contraction: both operands are read row-major).
// Workgroup = 128 (n) x 128 (c) tile of one tap over one slice of m; slices are combined with fp32 atomics.
// ---------------------------------------------------------------------------
struct WgradArgs {
  const float* dY; int lddy;     // [B*Lout, lddy]
  const float* X;  int ldx;      // [B*Lin, ldx]
  float* dW;                     // [ntaps][N][Cin] (row-major, reference layout after a permute), pre-zeroed
  int N, Cin, ntaps;
  int tap_d[MAX_TAPS];
  int M, Lgrid, Lin, Lout, in_stride, out_stride, out_off;
  int msplit;
  float* db;                     // optional (split kernel): bias gradient = column sums of dY, accumulated by the workgroups of tap 0 / c-tile 0
};

__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p, int tiles_n, int tiles_c) {
  constexpr int MK = 16, LDSW = 132;
  __shared__ __attribute__((aligned(16))) float ys[MK][LDSW];
  __shared__ __attribute__((aligned(16))) float xs[MK][LDSW];
  int bid = blockIdx.x;
  const int ms = bid % p.msplit; bid /= p.msplit;
  const int tc = bid % tiles_c; bid /= tiles_c;
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tap = bid;
  const int n0 = tn * 128, c0 = tc * 128, d = p.tap_d[tap];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave >> 1, wc = wave & 1;
  const int chunk = ((p.M + p.msplit - 1) / p.msplit + MK - 1) / MK * MK;
  const int m_begin = ms * chunk, m_end = min(p.M, m_begin + chunk);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const bool plain = (p.out_stride == 1 && p.out_off == 0 && p.Lgrid == p.Lout && p.in_stride == 1 && p.Lin == p.Lgrid);
  for (int mb = m_begin; mb < m_end; mb += MK) {
    // stage MK rows of dY (columns n0..) and of the tap-shifted X (columns c0..): 2 x float4 per thread per operand
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 256 * i, row = e >> 5, c4 = (e & 31) * 4;
      const int m = mb + row;
      f32x4 yv = {0.f, 0.f, 0.f, 0.f}, xv = {0.f, 0.f, 0.f, 0.f};
      if (m < m_end) {
        int orow = m, l, b;
        if (plain) { b = m / p.Lgrid; l = m - b * p.Lgrid; }
        else { b = m / p.Lgrid; l = m - b * p.Lgrid; orow = b * p.Lout + l * p.out_stride + p.out_off; }
        const int ls = l * p.in_stride + d;
        if (n0 + c4 < p.N) {
          const float* yp = p.dY + (size_t)orow * p.lddy + n0 + c4;
          if (n0 + c4 + 3 < p.N) yv = *reinterpret_cast<const f32x4*>(yp);
          else for (int k = 0; k < 4; ++k) if (n0 + c4 + k < p.N) yv[k] = yp[k];
        }
        if ((unsigned)ls < (unsigned)p.Lin && c0 + c4 < p.Cin) {
          const float* xp = p.X + (size_t)(b * p.Lin + ls) * p.ldx + c0 + c4;
          if (c0 + c4 + 3 < p.Cin) xv = *reinterpret_cast<const f32x4*>(xp);
          else for (int k = 0; k < 4; ++k) if (c0 + c4 + k < p.Cin) xv[k] = xp[k];
        }
      }
      *reinterpret_cast<f32x4*>(&ys[row][c4]) = yv;
      *reinterpret_cast<f32x4*>(&xs[row][c4]) = xv;
    }
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < MK / 2; ++k2) {
      const int mr = 2 * k2 + (lane >> 5);
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = ys[mr][wn * 64 + i * 32 + (lane & 31)];
        b[i] = xs[mr][wc * 64 + i * 32 + (lane & 31)];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D[i = n][j = c]: lane&31 = c, registers = n rows
  float* out = p.dW + (size_t)tap * p.N * p.Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + wc * 64 + j * 32 + (lane & 31);
      if (c >= p.Cin) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < p.N) atomicAdd(out + (size_t)n * p.Cin + c, acc[i][j][r]);
      }
    }
}

// ---------------------------------------------------------------------------
// The same weight gradient on the bf16 MFMA at fp32 accuracy (round 3; option split_bf16, as the forward tap-GEMM in
// gemm.hip: each fp32 operand = three bf16 terms, six products, smallest first).  The contraction index is the GEMM row m,
// which is the STRIDED index of both operands in memory, while a v_mfma_f32_32x32x16_bf16 lane wants 8 consecutive k.
// The transposition happens in registers: a thread loads a 4 (m) x 4 (columns) patch of each operand (four float4), so it
// holds 4 consecutive m for each of its 4 columns and writes them as 8-byte k-runs into the [column][k] LDS planes the
// forward kernel uses (80-byte rows, three planes per operand, 60 KB: two workgroups per CU).  Lanes 0-7 of every 8 are
// the eight 4-row groups of a K-step of 32 rows, so the 8-byte writes of one column row are contiguous (a wave's write
// instruction is 8 column rows x 64 bytes, 320 bytes apart: 4 lanes per bank, the 4 cycles that 512 bytes cost anyway).
// Workgroup = 128 (n) x 128 (c) tile of one tap over one slice of m, 4 waves 2 x 2, fp32 atomics across slices; global
// loads run two K-steps ahead in two register sets.
// ---------------------------------------------------------------------------
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wg_u32x2 __attribute__((ext_vector_type(2)));
constexpr int WG_ROW = 80, WG_PLANE = 256 * WG_ROW, WG_SMEM = 3 * WG_PLANE;

__device__ __forceinline__ void wg_split3(const f32x4 v, wg_u32x2& h, wg_u32x2& m, wg_u32x2& l) {
  const wg_bf16x4 b0 = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  f32x4 r1;
#pragma unroll
  for (int e = 0; e < 4; ++e) r1[e] = v[e] - (float)b0[e];
  const wg_bf16x4 b1 = {(__bf16)r1[0], (__bf16)r1[1], (__bf16)r1[2], (__bf16)r1[3]};
  f32x4 r2;
#pragma unroll
  for (int e = 0; e < 4; ++e) r2[e] = r1[e] - (float)b1[e];
  const wg_bf16x4 b2 = {(__bf16)r2[0], (__bf16)r2[1], (__bf16)r2[2], (__bf16)r2[3]};
  h = __builtin_bit_cast(wg_u32x2, b0);
  m = __builtin_bit_cast(wg_u32x2, b1);
  l = __builtin_bit_cast(wg_u32x2, b2);
}

__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(const WgradArgs p, int tiles_n, int tiles_c) {
  constexpr int MK = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
  int bid = blockIdx.x;
  const int ms = bid % p.msplit; bid /= p.msplit;
  const int tc = bid % tiles_c; bid /= tiles_c;
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tap = bid;
  const int n0 = tn * 128, c0 = tc * 128, d = p.tap_d[tap];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave >> 1, wc = wave & 1;
  const int mq = tid & 7, cq = tid >> 3;                       // rows 4 mq .. 4 mq + 3 of a K-step, columns 4 cq .. 4 cq + 3
  const int chunk = ((p.M + p.msplit - 1) / p.msplit + MK - 1) / MK * MK;
  const int m_begin = ms * chunk, m_end = min(p.M, m_begin + chunk);
  const int nk = m_end > m_begin ? (m_end - m_begin + MK - 1) / MK : 0;
  const bool plain = (p.out_stride == 1 && p.out_off == 0 && p.Lgrid == p.Lout && p.in_stride == 1 && p.Lin == p.Lgrid);
  const bool y_full = n0 + 4 * cq + 3 < p.N, x_full = c0 + 4 * cq + 3 < p.Cin;

  // (clip, position) of the thread's first row, advanced by 32 rows per K-step
  int m_cur = m_begin + 4 * mq;
  int b_cur = m_cur / p.Lgrid, l_cur = m_cur - b_cur * p.Lgrid;
  f32x4 ry[2][4], rx[2][4];
  auto gload = [&](const int set) {
    int b = b_cur, l = l_cur;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 yv = {0.f, 0.f, 0.f, 0.f}, xv = {0.f, 0.f, 0.f, 0.f};
      if (m_cur + j < m_end) {
        const int orow = plain ? m_cur + j : b * p.Lout + l * p.out_stride + p.out_off;
        const int ls = l * p.in_stride + d;
        const float* yp = p.dY + (size_t)orow * p.lddy + n0 + 4 * cq;
        if (y_full) yv = *reinterpret_cast<const f32x4*>(yp);
        else for (int k = 0; k < 4; ++k) if (n0 + 4 * cq + k < p.N) yv[k] = yp[k];
        if ((unsigned)ls < (unsigned)p.Lin) {
          const float* xp = p.X + (size_t)(b * p.Lin + ls) * p.ldx + c0 + 4 * cq;
          if (x_full) xv = *reinterpret_cast<const f32x4*>(xp);
          else for (int k = 0; k < 4; ++k) if (c0 + 4 * cq + k < p.Cin) xv[k] = xp[k];
        }
      }
      ry[set][j] = yv;
      rx[set][j] = xv;
      if (++l == p.Lgrid) { l = 0; ++b; }
    }
    m_cur += MK;
    l_cur += MK;
    while (l_cur >= p.Lgrid) { l_cur -= p.Lgrid; ++b_cur; }
  };
  // transpose in registers, split, store: column e of the patch -> LDS row 32 e + cq, bytes 8 mq .. 8 mq + 7 of each plane
  const bool do_db = p.db != nullptr && tap == 0 && tc == 0;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  auto lstore = [&](const int set) {
    if (do_db) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[e] += (ry[set][0][e] + ry[set][1][e]) + (ry[set][2][e] + ry[set][3][e]);
    }
#pragma unroll
    for (int op = 0; op < 2; ++op) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 v = op == 0 ? f32x4{ry[set][0][e], ry[set][1][e], ry[set][2][e], ry[set][3][e]}
                                : f32x4{rx[set][0][e], rx[set][1][e], rx[set][2][e], rx[set][3][e]};
        wg_u32x2 h, m, l;
        wg_split3(v, h, m, l);
        unsigned char* dst = wg_smem + (op * 128 + 4 * cq + e) * WG_ROW + 8 * mq;
        *reinterpret_cast<wg_u32x2*>(dst) = h;
        *reinterpret_cast<wg_u32x2*>(dst + WG_PLANE) = m;
        *reinterpret_cast<wg_u32x2*>(dst + 2 * WG_PLANE) = l;
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  if (nk > 0) {
    gload(0);
    lstore(0);
    __syncthreads();
    if (nk > 1) gload(1);
    if (nk > 2) gload(0);
    const int frag_off = (lane & 31) * WG_ROW + 16 * (lane >> 5);
    const unsigned char* Ys = wg_smem + (wn * 64) * WG_ROW + frag_off;
    const unsigned char* Xs = wg_smem + (128 + wc * 64) * WG_ROW + frag_off;
    auto stage = [&](const int it, const int nset, const bool store_next) {
#pragma unroll
      for (int u = 0; u < MK / 16; ++u) {
        wg_bf16x8 a[3][2], b[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            a[pl][i] = *reinterpret_cast<const wg_bf16x8*>(Ys + pl * WG_PLANE + i * 32 * WG_ROW + u * 32);
            b[pl][i] = *reinterpret_cast<const wg_bf16x8*>(Xs + pl * WG_PLANE + i * 32 * WG_ROW + u * 32);
          }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x16 c = acc[i][j];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);   // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
            acc[i][j] = c;
          }
      }
      __syncthreads();                      // every wave is done reading the stage
      if (store_next) {
        lstore(nset);
        __syncthreads();
        if (it + 3 < nk) gload(nset);
      }
    };
    int it = 0;
    for (; it + 2 < nk; it += 2) {
      stage(it, 1, true);
      stage(it + 1, 0, true);
    }
    if (it + 1 < nk) {
      stage(it, 1, true);
      stage(it + 1, 0, false);
    } else {
      stage(it, 1, false);
    }
  }
  if (do_db) {
    // the thread's 4 columns over its rows; the 8 row groups of a column quad are 8 adjacent lanes
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = cs[e];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      if (mq == 0 && n0 + 4 * cq + e < p.N) atomicAdd(p.db + n0 + 4 * cq + e, v);
    }
  }
  // D[i][j]: registers = dY columns n, lane & 31 = X columns c (contiguous: the atomics of an instruction cover 128 bytes)
  float* out = p.dW + (size_t)tap * p.N * p.Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + wc * 64 + j * 32 + (lane & 31);
      if (c >= p.Cin) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < p.N) atomicAdd(out + (size_t)n * p.Cin + c, acc[i][j][r]);
      }
    }
}

int launch_colsum(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* out, hipStream_t s);
// db (optional, pre-zeroed [N]): the bias gradient, column sums of dY.  The split kernel accumulates it from the dY tiles it
// stages anyway (the workgroups of tap 0 and c-tile 0); with the fp32-MFMA kernel it is the column-sum kernel's launch.
int launch_wgrad(const float* dY, int lddy, const float* X, int ldx, float* dW, float* db, int N, int Cin, int kind, int B, int Lin,
                 hipStream_t s) {
  PT_CHECK(dY && X && dW && N > 0 && Cin > 0 && B > 0 && Lin > 0, "wgrad: bad argument");
  PT_CHECK(lddy % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)dY | (uintptr_t)X) & 15) == 0, "wgrad: rows must be 16-byte aligned");
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.dY = dY; a.lddy = lddy; a.X = X; a.ldx = ldx; a.dW = dW; a.N = N; a.Cin = Cin;
  a.db = option(OPT_SPLIT_BF16) == 1 ? db : nullptr;
  a.Lin = Lin; a.in_stride = 1; a.out_stride = 1; a.out_off = 0;
  int launches = 1;
  switch (kind) {
    case 0: a.ntaps = 1; a.Lgrid = a.Lout = Lin; break;
    case 1: a.ntaps = 3; for (int k = 0; k < 3; ++k) a.tap_d[k] = k - 1; a.Lgrid = a.Lout = Lin; break;
    case 2: a.ntaps = 7; for (int k = 0; k < 7; ++k) a.tap_d[k] = k - 3; a.Lgrid = a.Lout = Lin; break;
    case 3: a.ntaps = 4; for (int k = 0; k < 4; ++k) a.tap_d[k] = k - 1; a.in_stride = 2; a.Lgrid = a.Lout = (Lin - 2) / 2 + 1; break;
    case 4: launches = 2; a.ntaps = 4; a.Lgrid = Lin; a.Lout = 2 * Lin; a.out_stride = 2; break;
    default: PT_CHECK(false, "wgrad: kind %d unknown", kind);
  }
  a.M = B * a.Lgrid;
  const int tiles_n = (N + 127) / 128, tiles_c = (Cin + 127) / 128;
  for (int par = 0; par < launches; ++par) {
    if (kind == 4) {
      // ConvTranspose1d k4 s2 p1: out[2q] = x[q] w1 + x[q-1] w3 ; out[2q+1] = x[q] w2 + x[q+1] w0.  Taps not of this
      // output parity get an offset that never lands in range, so they contribute nothing.
      const int far = 1 << 28;
      a.out_off = par;
      if (par == 0) { a.tap_d[0] = far; a.tap_d[1] = 0; a.tap_d[2] = far; a.tap_d[3] = -1; }
      else { a.tap_d[0] = 1; a.tap_d[1] = far; a.tap_d[2] = 0; a.tap_d[3] = far; }
    }
    const int base = tiles_n * tiles_c * a.ntaps;
    // slices of m: the fp32-MFMA kernel runs many small workgroups per CU (1024 or more in flight); the split kernel runs two
    // 60 KB workgroups per CU, so its slices fill one round of 512 (few tiles: every slice costs 16 K atomics per tile) or two
    const bool split = option(OPT_SPLIT_BF16) == 1;
    int msplit = split ? (base <= 32 ? 512 : 1024) / base : (1024 + base - 1) / base;
    const int maxsplit = (a.M + 63) / 64;
    if (msplit > maxsplit) msplit = maxsplit;
    if (msplit < 1) msplit = 1;
    a.msplit = msplit;
    if (split) {
      PT_ENSURE_LDS(wgrad_split_kernel, WG_SMEM);
      PT_LAUNCH(wgrad_split_kernel, dim3(base * msplit), dim3(256), WG_SMEM, s, a, tiles_n, tiles_c);
    } else {
      PT_LAUNCH(wgrad_kernel, dim3(base * msplit), dim3(256), 0, s, a, tiles_n, tiles_c);
    }
  }
  if (db && !a.db) return launch_colsum(dY, lddy, nullptr, 0, B * a.Lout, N, db, s);
  return 0;
}

// column sums: out[c] += sum_rows a[row][c] (* b[row][c] when b != null)   (bias / BatchNorm / LayerNorm-affine gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                     int ldb, int rows, int cols, float* __restrict__ out, int rows_per_block) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  double s = 0.0;
  if (c < cols)
    for (int r = r0 + sub; r < r1; r += 4) {
      const float v = a[(size_t)r * lda + c];
      s += b ? (double)v * (double)b[(size_t)r * ldb + c] : (double)v;
    }
  __shared__ double red[4][64];
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && c < cols) atomicAdd(out + c, (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]));
}
int launch_colsum(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* out, hipStream_t s) {
  PT_CHECK(a && out && rows > 0 && cols > 0, "colsum: bad argument");
  const int rpb = 256;
  PT_LAUNCH(colsum_kernel, dim3((cols + 63) / 64, (rows + rpb - 1) / rpb), dim3(256), 0, s, a, lda, b, ldb, rows, cols, out, rpb);
  return 0;
}

// ---------------------------------------------------------------------------
// LayerNorm over 512-wide rows, one wave per row.  fwd saves mean / rstd; bwd accumulates dgamma / dbeta.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                     const float* __restrict__ g, const float* __restrict__ b,
                                                     float* __restrict__ y, float* __restrict__ stats, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = (i >> 2) * 256 + lane * 4 + (i & 3);
    v[i] = x[(size_t)row * DM + c] + (res ? res[(size_t)row * DM + c] : 0.f);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  const float mean = wsum(s) * (1.f / DM);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) q += (v[i] - mean) * (v[i] - mean);
  const float rstd = 1.0f / sqrtf(wsum(q) * (1.f / DM) + 1e-5f);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = (i >> 2) * 256 + lane * 4 + (i & 3);
    y[(size_t)row * DM + c] = (v[i] - mean) * rstd * g[c] + b[c];
  }
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
// xin = the LayerNorm INPUT (x + res of the forward); dx = gradient w.r.t. that input
// Each wave owns a contiguous block of rows (two rows in flight per iteration); the gamma / beta partial sums of the
// workgroup's 8 waves are combined in LDS, so a column receives one atomic per workgroup (<= 512 per launch) instead of
// one per 16 rows (2,816 at 45,056 rows: the launch was bound by the L2 atomic chains, 310 us against ~60 us of traffic).
__global__ __launch_bounds__(512) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ res, const float* __restrict__ stats,
                                                     const float* __restrict__ g, float* __restrict__ dx,
                                                     float* __restrict__ dg, float* __restrict__ db, int rows, int rows_per_wave) {
  __shared__ float red[8][2 * DM];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = blockIdx.x * 8 + wv;
  f32x4 ag[2], ab[2], gv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    ag[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[h] = ag[h];
    gv[h] = *reinterpret_cast<const f32x4*>(g + h * 256 + lane * 4);
  }
  const int r0 = w * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
  for (int row = r0; row < r1; row += 2) {
    f32x4 xv[2][2], dv[2][2];
    float mean[2], rstd[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int rr = min(row + k, r1 - 1);
      mean[k] = stats[2 * rr];
      rstd[k] = stats[2 * rr + 1];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const size_t o = (size_t)rr * DM + h * 256 + lane * 4;
        xv[k][h] = *reinterpret_cast<const f32x4*>(x + o);
        if (res) xv[k][h] += *reinterpret_cast<const f32x4*>(res + o);
        dv[k][h] = *reinterpret_cast<const f32x4*>(dy + o);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (row + k >= r1) break;
      f32x4 xh[2], gd[2];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        xh[h] = (xv[k][h] - mean[k]) * rstd[k];
        gd[h] = dv[k][h] * gv[h];
        ag[h] += dv[k][h] * xh[h];
        ab[h] += dv[k][h];
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1 += gd[h][e]; s2 += gd[h][e] * xh[h][e]; }
      }
      s1 = wsum(s1) * (1.f / DM);
      s2 = wsum(s2) * (1.f / DM);
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<f32x4*>(dx + (size_t)(row + k) * DM + h * 256 + lane * 4) = rstd[k] * (gd[h] - s1 - xh[h] * s2);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    *reinterpret_cast<f32x4*>(&red[wv][h * 256 + lane * 4]) = ag[h];
    *reinterpret_cast<f32x4*>(&red[wv][DM + h * 256 + lane * 4]) = ab[h];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * DM; c += 512) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][c];
    atomicAdd(c < DM ? dg + c : db + (c - DM), t);
  }
}
int launch_ln_fwd(const float* x, const float* res, const float* g, const float* b, float* y, float* stats, int rows,
                  hipStream_t s) {
  PT_CHECK(x && g && b && y && stats && rows > 0, "ln_fwd: bad argument");
  PT_LAUNCH(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, res, g, b, y, stats, rows);
  return 0;
}
int launch_ln_bwd(const float* dy, const float* x, const float* res, const float* stats, const float* g, float* dx,
                  float* dg, float* db, int rows, hipStream_t s) {
  PT_CHECK(dy && x && stats && g && dx && dg && db && rows > 0, "ln_bwd: bad argument");
  PT_CHECK(((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)res | (uintptr_t)g | (uintptr_t)dx) & 15) == 0), "ln_bwd: operands must be 16-byte aligned");
  const int rpw = std::max(8, (rows + 4095) / 4096), waves = (rows + rpw - 1) / rpw;       // at most 512 workgroups
  PT_LAUNCH(ln_bwd_kernel, dim3((waves + 7) / 8), dim3(512), 0, s, dy, x, res, stats, g, dx, dg, db, rows, rpw);
  return 0;
}

// ---------------------------------------------------------------------------
// stylization modulate + SiLU (PTransformer.py:134-139): h = SiLU(y * (1 + scale[c]) + shift[c]), c = label of the row's clip.
// table [n_classes][1024] = scale | shift.  bwd: dy, dtable (per-class sums, flushed when the class changes).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void modsilu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ table,
                                                          const int64_t* __restrict__ label, int n_label, int rows_per_clip,
                                                          int n_classes, float* __restrict__ h, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  long long lb = label[n_label == 1 ? 0 : row / rows_per_clip];
  lb = lb < 0 ? 0 : (lb >= n_classes ? n_classes - 1 : lb);
  const float* t = table + (size_t)lb * 2 * DM;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = (i >> 2) * 256 + lane * 4 + (i & 3);
    const float u = y[(size_t)row * DM + c] * (1.f + t[c]) + t[DM + c];
    h[(size_t)row * DM + c] = u / (1.f + expf(-u));
  }
}
// Per-class table gradients: register partials per wave, flushed into an LDS table (4 classes at most per workgroup
// image, LDS atomics) when the class changes, one global atomic per touched entry per workgroup at the end.
constexpr int MS_MAXC = 4;
__global__ __launch_bounds__(512) void modsilu_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ y,
                                                          const float* __restrict__ table, const int64_t* __restrict__ label,
                                                          int n_label, int rows_per_clip, int n_classes,
                                                          float* __restrict__ dy, float* __restrict__ dtable, int rows,
                                                          int rows_per_wave) {
  __shared__ float acc[MS_MAXC][2 * DM];
  const int lds_classes = n_classes <= MS_MAXC ? n_classes : 0;       // more classes: straight to global memory
  for (int k = threadIdx.x; k < lds_classes * 2 * DM; k += 512) (&acc[0][0])[k] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 8 + (threadIdx.x >> 6);
  const int r0 = w * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
  f32x4 asc[2], ash[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) { asc[h] = f32x4{0.f, 0.f, 0.f, 0.f}; ash[h] = asc[h]; }
  long long cur = -1;
  auto flush = [&]() {
    if (cur < 0) return;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = h * 256 + lane * 4 + e;
        if (lds_classes) {
          atomicAdd(&acc[cur][c], asc[h][e]);
          atomicAdd(&acc[cur][DM + c], ash[h][e]);
        } else {
          atomicAdd(dtable + (size_t)cur * 2 * DM + c, asc[h][e]);
          atomicAdd(dtable + (size_t)cur * 2 * DM + DM + c, ash[h][e]);
        }
      }
#pragma unroll
    for (int h = 0; h < 2; ++h) { asc[h] = f32x4{0.f, 0.f, 0.f, 0.f}; ash[h] = asc[h]; }
  };
  f32x4 sc[2], sh[2];
  for (int row = r0; row < r1; ++row) {
    long long lb = label[n_label == 1 ? 0 : row / rows_per_clip];
    lb = lb < 0 ? 0 : (lb >= n_classes ? n_classes - 1 : lb);
    if (lb != cur) {
      flush();
      cur = lb;
      const float* t = table + (size_t)lb * 2 * DM;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        sc[h] = *reinterpret_cast<const f32x4*>(t + h * 256 + lane * 4) + 1.f;
        sh[h] = *reinterpret_cast<const f32x4*>(t + DM + h * 256 + lane * 4);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const size_t o = (size_t)row * DM + h * 256 + lane * 4;
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o), dv = *reinterpret_cast<const f32x4*>(dh + o);
      f32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = yv[e] * sc[h][e] + sh[h][e];
        const float sg = 1.f / (1.f + expf(-u));
        const float du = dv[e] * (sg * (1.f + u * (1.f - sg)));
        out[e] = du * sc[h][e];
        asc[h][e] += du * yv[e];
        ash[h][e] += du;
      }
      *reinterpret_cast<f32x4*>(dy + o) = out;
    }
  }
  flush();
  __syncthreads();
  for (int k = threadIdx.x; k < lds_classes * 2 * DM; k += 512) {
    const float v = (&acc[0][0])[k];
    if (v != 0.f) atomicAdd(dtable + k, v);
  }
}
int launch_modsilu_fwd(const float* y, const float* table, const int64_t* label, int n_label, int rows_per_clip,
                       int n_classes, float* h, int rows, hipStream_t s) {
  PT_CHECK(y && table && label && h && rows > 0 && rows_per_clip > 0, "modsilu_fwd: bad argument");
  PT_LAUNCH(modsilu_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, y, table, label, n_label, rows_per_clip, n_classes, h, rows);
  return 0;
}
int launch_modsilu_bwd(const float* dh, const float* y, const float* table, const int64_t* label, int n_label,
                       int rows_per_clip, int n_classes, float* dy, float* dtable, int rows, hipStream_t s) {
  PT_CHECK(dh && y && table && label && dy && dtable && rows > 0 && rows_per_clip > 0, "modsilu_bwd: bad argument");
  PT_CHECK(((((uintptr_t)dh | (uintptr_t)y | (uintptr_t)table | (uintptr_t)dy) & 15) == 0), "modsilu_bwd: operands must be 16-byte aligned");
  const int rpw = std::max(8, (rows + 4095) / 4096), waves = (rows + rpw - 1) / rpw;       // at most 512 workgroups
  PT_LAUNCH(modsilu_bwd_kernel, dim3((waves + 7) / 8), dim3(512), 0, s, dh, y, table, label, n_label, rows_per_clip,
            n_classes, dy, dtable, rows, rpw);
  return 0;
}

// ---------------------------------------------------------------------------
// attention backward (head dim 64, no mask, optional dropout-free): recomputes the probabilities.
//   pass Q: one wave per query row -> dQ row, lse, delta = dO . O
//   pass KV: one wave per key row  -> dK, dV rows (no atomics)
// rows are b*N + i, head h at column h*64 (same addressing as pt_attention_cl).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                         int ldk, const float* __restrict__ V, int ldv,
                                                         const float* __restrict__ dO, int ldo, float* __restrict__ dQ,
                                                         int lddq, float* __restrict__ lse, float* __restrict__ delta,
                                                         int B, int Nq, int Nk, int H, float scale, float pdrop,
                                                         unsigned long long seed) {
  extern __shared__ float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* sc = sm + wave * 2 * Nk;     // scores, then probabilities
  float* dp = sc + Nk;
  const int gw = blockIdx.x * 4 + wave;
  if (gw >= B * H * Nq) return;
  const int i = gw % Nq, h = (gw / Nq) % H, b = gw / (Nq * H);
  const float q = Q[(size_t)(b * Nq + i) * ldq + h * 64 + lane];
  const float go = dO[(size_t)(b * Nq + i) * ldo + h * 64 + lane];
  const unsigned thr = (unsigned)((double)pdrop * 4294967296.0);
  const float ks = 1.f / (1.f - pdrop);
  const size_t mbase = ((size_t)(b * H + h) * Nq + i) * Nk;
  float mx = -INFINITY;
  for (int j = 0; j < Nk; ++j) {
    const float s = wsum(q * K[(size_t)(b * Nk + j) * ldk + h * 64 + lane]) * scale;
    float d = wsum(go * V[(size_t)(b * Nk + j) * ldv + h * 64 + lane]);
    if (pdrop > 0.f) d *= attn_keep(seed, mbase + j, thr, ks);     // dA = dAd * mask / (1-p)
    if (lane == 0) { sc[j] = s; dp[j] = d; }
    mx = fmaxf(mx, s);
  }
  float l = 0.f;
  for (int j = lane; j < Nk; j += 64) l += expf(sc[j] - mx);
  l = wsum(l);
  const float ls = mx + logf(l);
  float dl = 0.f;
  for (int j = lane; j < Nk; j += 64) dl += expf(sc[j] - ls) * dp[j];
  dl = wsum(dl);
  float dq = 0.f;
  for (int j = 0; j < Nk; ++j) {
    const float pj = expf(sc[j] - ls);
    dq += pj * (dp[j] - dl) * scale * K[(size_t)(b * Nk + j) * ldk + h * 64 + lane];
  }
  dQ[(size_t)(b * Nq + i) * lddq + h * 64 + lane] = dq;
  if (lane == 0) { lse[(size_t)(b * H + h) * Nq + i] = ls; delta[(size_t)(b * H + h) * Nq + i] = dl; }
}
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                          int ldk, const float* __restrict__ V, int ldv,
                                                          const float* __restrict__ dO, int ldo, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, float* __restrict__ dK, int lddk,
                                                          float* __restrict__ dV, int lddv, int B, int Nq, int Nk, int H,
                                                          float scale, float pdrop, unsigned long long seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + wave;
  if (gw >= B * H * Nk) return;
  const int j = gw % Nk, h = (gw / Nk) % H, b = gw / (Nk * H);
  const float k = K[(size_t)(b * Nk + j) * ldk + h * 64 + lane];
  const float v = V[(size_t)(b * Nk + j) * ldv + h * 64 + lane];
  float dk = 0.f, dv = 0.f;
  const unsigned thr = (unsigned)((double)pdrop * 4294967296.0);
  const float ks = 1.f / (1.f - pdrop);
  for (int i = 0; i < Nq; ++i) {
    const float q = Q[(size_t)(b * Nq + i) * ldq + h * 64 + lane];
    const float go = dO[(size_t)(b * Nq + i) * ldo + h * 64 + lane];
    const float s = wsum(q * k) * scale;
    float d = wsum(go * v);
    const float p = expf(s - lse[(size_t)(b * H + h) * Nq + i]);
    const float keep = pdrop > 0.f ? attn_keep(seed, ((size_t)(b * H + h) * Nq + i) * Nk + j, thr, ks) : 1.f;
    d *= keep;
    dv += p * keep * go;
    dk += p * (d - delta[(size_t)(b * H + h) * Nq + i]) * scale * q;
  }
  dK[(size_t)(b * Nk + j) * lddk + h * 64 + lane] = dk;
  dV[(size_t)(b * Nk + j) * lddv + h * 64 + lane] = dv;
}
// forward with dropout on the attention weights (nn.MultiheadAttention(dropout=0.1) in train mode): one wave per query row
__global__ __launch_bounds__(256) void attn_fwd_train_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                             int ldk, const float* __restrict__ V, int ldv, float* __restrict__ O,
                                                             int ldo, int B, int Nq, int Nk, int H, float scale, float pdrop,
                                                             unsigned long long seed) {
  extern __shared__ float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* sc = sm + wave * Nk;
  const int gw = blockIdx.x * 4 + wave;
  if (gw >= B * H * Nq) return;
  const int i = gw % Nq, h = (gw / Nq) % H, b = gw / (Nq * H);
  const float q = Q[(size_t)(b * Nq + i) * ldq + h * 64 + lane];
  float mx = -INFINITY;
  for (int j = 0; j < Nk; ++j) {
    const float s = wsum(q * K[(size_t)(b * Nk + j) * ldk + h * 64 + lane]) * scale;
    if (lane == 0) sc[j] = s;
    mx = fmaxf(mx, s);
  }
  float l = 0.f;
  for (int j = lane; j < Nk; j += 64) l += expf(sc[j] - mx);
  l = wsum(l);
  const unsigned thr = (unsigned)((double)pdrop * 4294967296.0);
  const float ks = 1.f / (1.f - pdrop);
  const size_t mbase = ((size_t)(b * H + h) * Nq + i) * Nk;
  float o = 0.f;
  for (int j = 0; j < Nk; ++j)
    o += expf(sc[j] - mx) / l * attn_keep(seed, mbase + j, thr, ks) * V[(size_t)(b * Nk + j) * ldv + h * 64 + lane];
  O[(size_t)(b * Nq + i) * ldo + h * 64 + lane] = o;
}
// ---------------------------------------------------------------------------
// Tiled attention for training (fp32, head dim 64): one 512-thread workgroup per (clip, head) keeps Q, K, V (and dO) of
// that head and the Nq x Nk score / probability matrices in LDS; every contraction is a register-tiled 4x4 FMA GEMM
// whose operands are read from LDS as float4, in their natural row-major layout (no transposes):
//   "kc"  C[m][n] = sum_k A[m][k] B[n][k]   (k contiguous in both: S = Q K^T, dPd = dO V^T)
//   "km"  C[m][n] = sum_k A[k][m] B[k][n]   (k-major in both:      dV = Pd^T dO, dK = dS^T Q)
//   "mix" C[m][n] = sum_k A[m][k] B[k][n]   (O = Pd V, dQ = dS K)
// fp32 MFMA has the same peak as the fp32 vector unit on gfx950 (157 TFLOP/s), so the vector FMAs cost nothing here and
// keep the summation free of layout shuffles.  LDS need at Nq = Nk = 88 (the predictor's 22 x 4 tokens): 158 KB of the
// 160 KB, one workgroup per CU.  Longer sequences take the query-blocked variants below (K and V resident, up to 212 keys in
// the backward pass); what fits neither falls back to the one-wave-per-row kernels above.
// The dropout mask index is the same ((b*H+h)*Nq+i)*Nk+j as in those kernels.
// ---------------------------------------------------------------------------
constexpr int AT = 512;
__host__ __device__ inline int at_ru4(int n) { return (n + 3) & ~3; }
__host__ __device__ inline int at_pitch(int n) { int p = at_ru4(n); return (p & 7) == 4 ? p : p + 4; }   // odd multiple of 4

__device__ __forceinline__ void at_load_rows(float* dst, int pitch, const float* __restrict__ src, int ld, int n, int npad) {
  for (int e = threadIdx.x; e < npad * 16; e += AT) {
    const int r = e >> 4, c = (e & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n) v = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c);
    *reinterpret_cast<float4*>(dst + r * pitch + c) = v;
  }
}
// kc tile: rows m = mi + a*TM (interleaved: consecutive lanes hit consecutive LDS rows), n = n0 + c, K = 64
__device__ __forceinline__ void at_kc(const float* A, int lda, const float* B, int ldb, int mi, int TM, int n0, float (&acc)[4][4]) {
#pragma unroll 4
  for (int k = 0; k < 64; k += 4) {
    float4 a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const float4*>(A + (mi + r * TM) * lda + k);
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4*>(B + (n0 + c) * ldb + k);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[r][c] += a[r].x * b[c].x + a[r].y * b[c].y + a[r].z * b[c].z + a[r].w * b[c].w;
  }
}
// km tile: C[m0+r][n0+c] += sum_{k<K} A[k][m0+r] * B[k][n0+c]
__device__ __forceinline__ void at_km(const float* A, int lda, const float* B, int ldb, int K, int m0, int n0, float (&acc)[4][4]) {
#pragma unroll 4
  for (int k = 0; k < K; ++k) {
    const float4 a = *reinterpret_cast<const float4*>(A + k * lda + m0);
    const float4 b = *reinterpret_cast<const float4*>(B + k * ldb + n0);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] += av[r] * bv[c];
  }
}
// mix tile: C[m0+r][n0+c] += sum_{k<K4} A[m0+r][k] * B[k][n0+c]   (K4 multiple of 4, pads zero)
__device__ __forceinline__ void at_mix(const float* A, int lda, const float* B, int ldb, int K4, int m0, int n0, float (&acc)[4][4]) {
#pragma unroll 2
  for (int k = 0; k < K4; k += 4) {
    float4 a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const float4*>(A + (m0 + r) * lda + k);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b[kk] = *reinterpret_cast<const float4*>(B + (k + kk) * ldb + n0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[r][0] += a[r].x * b[0].x + a[r].y * b[1].x + a[r].z * b[2].x + a[r].w * b[3].x;
      acc[r][1] += a[r].x * b[0].y + a[r].y * b[1].y + a[r].z * b[2].y + a[r].w * b[3].y;
      acc[r][2] += a[r].x * b[0].z + a[r].y * b[1].z + a[r].z * b[2].z + a[r].w * b[3].z;
      acc[r][3] += a[r].x * b[0].w + a[r].y * b[1].w + a[r].z * b[2].w + a[r].w * b[3].w;
    }
  }
}
__device__ __forceinline__ void at_store_rows(float* __restrict__ dst, int ld, int m0, int n0, int nrows, const float (&acc)[4][4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (m0 + r < nrows)
      *reinterpret_cast<float4*>(dst + (size_t)(m0 + r) * ld + n0) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
}

__global__ __launch_bounds__(AT) void attn_fwd_tiled_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                            int ldk, const float* __restrict__ V, int ldv, float* __restrict__ O,
                                                            int ldo, int Nq, int Nk, int H, float scale, float pdrop,
                                                            unsigned long long seed) {
  extern __shared__ __attribute__((aligned(16))) float at_sm[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int NQ4 = at_ru4(Nq), NK4 = at_ru4(Nk), pp = at_pitch(Nk);
  float* Qs = at_sm;                  // [NQ4][68]
  float* Ks = Qs + NQ4 * 68;          // [NK4][64]
  float* Vs = Ks + NK4 * 64;          // [NK4][64]
  float* Ps = Vs + NK4 * 64;          // [NQ4][pp]
  at_load_rows(Qs, 68, Q + (size_t)b * Nq * ldq + h * 64, ldq, Nq, NQ4);
  at_load_rows(Ks, 64, K + (size_t)b * Nk * ldk + h * 64, ldk, Nk, NK4);
  at_load_rows(Vs, 64, V + (size_t)b * Nk * ldv + h * 64, ldv, Nk, NK4);
  __syncthreads();
  const int TM = NQ4 / 4, TN = NK4 / 4;
  for (int t = threadIdx.x; t < TM * TN; t += AT) {
    const int mi = t % TM, n0 = (t / TM) * 4;
    float acc[4][4] = {};
    at_kc(Qs, 68, Ks, 64, mi, TM, n0, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4*>(Ps + (mi + r * TM) * pp + n0) =
          make_float4(acc[r][0] * scale, acc[r][1] * scale, acc[r][2] * scale, acc[r][3] * scale);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned thr = (unsigned)((double)pdrop * 4294967296.0);
  const float ks = 1.f / (1.f - pdrop);
  for (int i = wave; i < NQ4; i += AT / 64) {
    float* row = Ps + i * pp;
    if (i >= Nq) {
      for (int j = lane; j < NK4; j += 64) row[j] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int j = lane; j < Nk; j += 64) mx = fmaxf(mx, row[j]);
    mx = wmax(mx);
    float l = 0.f;
    for (int j = lane; j < Nk; j += 64) l += expf(row[j] - mx);
    l = wsum(l);
    const size_t mbase = ((size_t)(b * H + h) * Nq + i) * Nk;
    for (int j = lane; j < NK4; j += 64) {
      float pj = 0.f;
      if (j < Nk) {
        pj = expf(row[j] - mx) / l;
        if (pdrop > 0.f) pj *= attn_keep(seed, mbase + j, thr, ks);
      }
      row[j] = pj;
    }
  }
  __syncthreads();
  float* Ob = O + (size_t)b * Nq * ldo + h * 64;
  for (int t = threadIdx.x; t < TM * 16; t += AT) {
    const int n0 = (t & 15) * 4, m0 = (t >> 4) * 4;
    float acc[4][4] = {};
    at_mix(Ps, pp, Vs, 64, NK4, m0, n0, acc);
    at_store_rows(Ob, ldo, m0, n0, Nq, acc);
  }
}

template <int JMAX>
__global__ __launch_bounds__(AT) void attn_bwd_tiled_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                            int ldk, const float* __restrict__ V, int ldv,
                                                            const float* __restrict__ dO, int ldo, float* __restrict__ dQ,
                                                            int lddq, float* __restrict__ dK, int lddk, float* __restrict__ dV,
                                                            int lddv, int Nq, int Nk, int H, float scale, float pdrop,
                                                            unsigned long long seed) {
  extern __shared__ __attribute__((aligned(16))) float at_sm[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int NQ4 = at_ru4(Nq), NK4 = at_ru4(Nk), pp = at_pitch(Nk);
  float* Qs = at_sm;                  // [NQ4][68]
  float* Gs = Qs + NQ4 * 68;          // dO [NQ4][68]
  float* Ks = Gs + NQ4 * 68;          // [NK4][64]
  float* Vs = Ks + NK4 * 64;          // [NK4][64]
  float* Ps = Vs + NK4 * 64;          // S, then Pd = P * keep        [NQ4][pp]
  float* Ds = Ps + NQ4 * pp;          // dPd, then dS                 [NQ4][pp]
  at_load_rows(Qs, 68, Q + (size_t)b * Nq * ldq + h * 64, ldq, Nq, NQ4);
  at_load_rows(Gs, 68, dO + (size_t)b * Nq * ldo + h * 64, ldo, Nq, NQ4);
  at_load_rows(Ks, 64, K + (size_t)b * Nk * ldk + h * 64, ldk, Nk, NK4);
  at_load_rows(Vs, 64, V + (size_t)b * Nk * ldv + h * 64, ldv, Nk, NK4);
  __syncthreads();
  const int TM = NQ4 / 4, TN = NK4 / 4;
  for (int t = threadIdx.x; t < TM * TN; t += AT) {
    const int mi = t % TM, n0 = (t / TM) * 4;
    float acc[4][4] = {}, acd[4][4] = {};
    at_kc(Qs, 68, Ks, 64, mi, TM, n0, acc);
    at_kc(Gs, 68, Vs, 64, mi, TM, n0, acd);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      *reinterpret_cast<float4*>(Ps + (mi + r * TM) * pp + n0) =
          make_float4(acc[r][0] * scale, acc[r][1] * scale, acc[r][2] * scale, acc[r][3] * scale);
      *reinterpret_cast<float4*>(Ds + (mi + r * TM) * pp + n0) = make_float4(acd[r][0], acd[r][1], acd[r][2], acd[r][3]);
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned thr = (unsigned)((double)pdrop * 4294967296.0);
  const float ks = 1.f / (1.f - pdrop);
  for (int i = wave; i < NQ4; i += AT / 64) {
    float* prow = Ps + i * pp;
    float* drow = Ds + i * pp;
    if (i >= Nq) {
      for (int j = lane; j < NK4; j += 64) { prow[j] = 0.f; drow[j] = 0.f; }
      continue;
    }
    float sv[JMAX], mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < JMAX; ++u) {
      const int j = lane + 64 * u;
      sv[u] = j < Nk ? prow[j] : -INFINITY;
      mx = fmaxf(mx, sv[u]);
    }
    mx = wmax(mx);
    float l = 0.f;
#pragma unroll
    for (int u = 0; u < JMAX; ++u) {
      sv[u] = (lane + 64 * u) < Nk ? expf(sv[u] - mx) : 0.f;
      l += sv[u];
    }
    l = wsum(l);
    const size_t mbase = ((size_t)(b * H + h) * Nq + i) * Nk;
    float kp[JMAX], dp[JMAX], dl = 0.f;
#pragma unroll
    for (int u = 0; u < JMAX; ++u) {
      const int j = lane + 64 * u;
      sv[u] = sv[u] / l;                                        // P
      kp[u] = (pdrop > 0.f && j < Nk) ? attn_keep(seed, mbase + j, thr, ks) : 1.f;
      dp[u] = j < Nk ? drow[j] * kp[u] : 0.f;                   // dP = dPd * keep
      dl += sv[u] * dp[u];
    }
    dl = wsum(dl);
#pragma unroll
    for (int u = 0; u < JMAX; ++u) {
      const int j = lane + 64 * u;
      if (j < NK4) {
        prow[j] = j < Nk ? sv[u] * kp[u] : 0.f;                 // Pd
        drow[j] = j < Nk ? sv[u] * (dp[u] - dl) * scale : 0.f;  // dS (scale folded)
      }
    }
  }
  __syncthreads();
  float* dKb = dK + (size_t)b * Nk * lddk + h * 64;
  float* dVb = dV + (size_t)b * Nk * lddv + h * 64;
  float* dQb = dQ + (size_t)b * Nq * lddq + h * 64;
  const int nkv = TN * 16, nq = TM * 16;
  for (int t = threadIdx.x; t < nkv + nq; t += AT) {
    if (t < nkv) {
      const int n0 = (t & 15) * 4, m0 = (t >> 4) * 4;
      float av[4][4] = {}, ak[4][4] = {};
      at_km(Ps, pp, Gs, 68, Nq, m0, n0, av);                    // dV = Pd^T dO
      at_km(Ds, pp, Qs, 68, Nq, m0, n0, ak);                    // dK = dS^T Q
      at_store_rows(dVb, lddv, m0, n0, Nk, av);
      at_store_rows(dKb, lddk, m0, n0, Nk, ak);
    } else {
      const int u = t - nkv, n0 = (u & 15) * 4, m0 = (u >> 4) * 4;
      float aq[4][4] = {};
      at_mix(Ds, pp, Ks, 64, NK4, m0, n0, aq);                  // dQ = dS K
      at_store_rows(dQb, lddq, m0, n0, Nq, aq);
    }
  }
}

// ---- query-blocked variants (round 3): K and V of a (clip, head) stay in LDS, the queries pass through in blocks, so the
// refiner's 180-frame train step (Nq = Nk = 180: K + V = 92 KB) no longer falls back to the one-wave-per-row kernels.
// Forward: grid (clip x head, query block), 64 queries per block.  Backward: ONE workgroup per (clip, head) walks the
// query blocks (32 queries each), writes dQ block by block and keeps its dK / dV tiles in registers across the blocks:
// no atomics, the summation order over queries is fixed.  Same tile routines, same dropout mask index as above.
constexpr int AT_QF = 64, AT_QB = 32;

__global__ __launch_bounds__(AT) void attn_fwd_qblock_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                             int ldk, const float* __restrict__ V, int ldv, float* __restrict__ O,
                                                             int ldo, int Nq, int Nk, int H, float scale, float pdrop,
                                                             unsigned long long seed) {
  extern __shared__ __attribute__((aligned(16))) float at_sm[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int q0 = blockIdx.y * AT_QF;
  const int nq = min(AT_QF, Nq - q0);                 // live queries of this block
  const int NQ4 = at_ru4(nq), NK4 = at_ru4(Nk), pp = at_pitch(Nk);
  float* Qs = at_sm;                  // [AT_QF][68]
  float* Ks = Qs + AT_QF * 68;        // [NK4][64]
  float* Vs = Ks + NK4 * 64;          // [NK4][64]
  float* Ps = Vs + NK4 * 64;          // [AT_QF][pp]
  at_load_rows(Qs, 68, Q + (size_t)(b * Nq + q0) * ldq + h * 64, ldq, nq, NQ4);
  at_load_rows(Ks, 64, K + (size_t)b * Nk * ldk + h * 64, ldk, Nk, NK4);
  at_load_rows(Vs, 64, V + (size_t)b * Nk * ldv + h * 64, ldv, Nk, NK4);
  __syncthreads();
  const int TM = NQ4 / 4, TN = NK4 / 4;
  for (int t = threadIdx.x; t < TM * TN; t += AT) {
    const int mi = t % TM, n0 = (t / TM) * 4;
    float acc[4][4] = {};
    at_kc(Qs, 68, Ks, 64, mi, TM, n0, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4*>(Ps + (mi + r * TM) * pp + n0) =
          make_float4(acc[r][0] * scale, acc[r][1] * scale, acc[r][2] * scale, acc[r][3] * scale);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned thr = (unsigned)((double)pdrop * 4294967296.0);
  const float ks = 1.f / (1.f - pdrop);
  for (int i = wave; i < NQ4; i += AT / 64) {
    float* row = Ps + i * pp;
    if (i >= nq) {
      for (int j = lane; j < NK4; j += 64) row[j] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int j = lane; j < Nk; j += 64) mx = fmaxf(mx, row[j]);
    mx = wmax(mx);
    float l = 0.f;
    for (int j = lane; j < Nk; j += 64) l += expf(row[j] - mx);
    l = wsum(l);
    const size_t mbase = ((size_t)(b * H + h) * Nq + q0 + i) * Nk;
    for (int j = lane; j < NK4; j += 64) {
      float pj = 0.f;
      if (j < Nk) {
        pj = expf(row[j] - mx) / l;
        if (pdrop > 0.f) pj *= attn_keep(seed, mbase + j, thr, ks);
      }
      row[j] = pj;
    }
  }
  __syncthreads();
  float* Ob = O + (size_t)(b * Nq + q0) * ldo + h * 64;
  for (int t = threadIdx.x; t < TM * 16; t += AT) {
    const int n0 = (t & 15) * 4, m0 = (t >> 4) * 4;
    float acc[4][4] = {};
    at_mix(Ps, pp, Vs, 64, NK4, m0, n0, acc);
    at_store_rows(Ob, ldo, m0, n0, nq, acc);
  }
}

// JMAX = ceil(Nk / 64) score columns per lane; KT = dK / dV tiles (4 keys x 4 columns) per thread: ceil(Nk/4 * 16 / AT)
template <int JMAX, int KT>
__global__ __launch_bounds__(AT) void attn_bwd_qblock_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                             int ldk, const float* __restrict__ V, int ldv,
                                                             const float* __restrict__ dO, int ldo, float* __restrict__ dQ,
                                                             int lddq, float* __restrict__ dK, int lddk, float* __restrict__ dV,
                                                             int lddv, int Nq, int Nk, int H, float scale, float pdrop,
                                                             unsigned long long seed) {
  extern __shared__ __attribute__((aligned(16))) float at_sm[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int NK4 = at_ru4(Nk), pp = at_pitch(Nk), TN = NK4 / 4;
  float* Qs = at_sm;                  // [AT_QB][68]
  float* Gs = Qs + AT_QB * 68;        // dO [AT_QB][68]
  float* Ks = Gs + AT_QB * 68;        // [NK4][64]
  float* Vs = Ks + NK4 * 64;          // [NK4][64]
  float* Ps = Vs + NK4 * 64;          // S, then Pd = P * keep        [AT_QB][pp]
  float* Ds = Ps + AT_QB * pp;        // dPd, then dS                 [AT_QB][pp]
  at_load_rows(Ks, 64, K + (size_t)b * Nk * ldk + h * 64, ldk, Nk, NK4);
  at_load_rows(Vs, 64, V + (size_t)b * Nk * ldv + h * 64, ldv, Nk, NK4);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned thr = (unsigned)((double)pdrop * 4294967296.0);
  const float ks = 1.f / (1.f - pdrop);
  const int nkv = TN * 16;            // dK / dV tiles of this head
  float av[KT][4][4] = {}, ak[KT][4][4] = {};
  for (int q0 = 0; q0 < Nq; q0 += AT_QB) {
    const int nq = min(AT_QB, Nq - q0), NQ4 = at_ru4(nq), TM = NQ4 / 4;
    __syncthreads();                  // the previous block's tiles are read (and K / V have landed, first time round)
    at_load_rows(Qs, 68, Q + (size_t)(b * Nq + q0) * ldq + h * 64, ldq, nq, NQ4);
    at_load_rows(Gs, 68, dO + (size_t)(b * Nq + q0) * ldo + h * 64, ldo, nq, NQ4);
    __syncthreads();
    for (int t = threadIdx.x; t < TM * TN; t += AT) {
      const int mi = t % TM, n0 = (t / TM) * 4;
      float acc[4][4] = {}, acd[4][4] = {};
      at_kc(Qs, 68, Ks, 64, mi, TM, n0, acc);
      at_kc(Gs, 68, Vs, 64, mi, TM, n0, acd);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        *reinterpret_cast<float4*>(Ps + (mi + r * TM) * pp + n0) =
            make_float4(acc[r][0] * scale, acc[r][1] * scale, acc[r][2] * scale, acc[r][3] * scale);
        *reinterpret_cast<float4*>(Ds + (mi + r * TM) * pp + n0) = make_float4(acd[r][0], acd[r][1], acd[r][2], acd[r][3]);
      }
    }
    __syncthreads();
    for (int i = wave; i < NQ4; i += AT / 64) {
      float* prow = Ps + i * pp;
      float* drow = Ds + i * pp;
      if (i >= nq) {
        for (int j = lane; j < NK4; j += 64) { prow[j] = 0.f; drow[j] = 0.f; }
        continue;
      }
      float sv[JMAX], mx = -INFINITY;
#pragma unroll
      for (int u = 0; u < JMAX; ++u) {
        const int j = lane + 64 * u;
        sv[u] = j < Nk ? prow[j] : -INFINITY;
        mx = fmaxf(mx, sv[u]);
      }
      mx = wmax(mx);
      float l = 0.f;
#pragma unroll
      for (int u = 0; u < JMAX; ++u) {
        sv[u] = (lane + 64 * u) < Nk ? expf(sv[u] - mx) : 0.f;
        l += sv[u];
      }
      l = wsum(l);
      const size_t mbase = ((size_t)(b * H + h) * Nq + q0 + i) * Nk;
      float kp[JMAX], dp[JMAX], dl = 0.f;
#pragma unroll
      for (int u = 0; u < JMAX; ++u) {
        const int j = lane + 64 * u;
        sv[u] = sv[u] / l;                                        // P
        kp[u] = (pdrop > 0.f && j < Nk) ? attn_keep(seed, mbase + j, thr, ks) : 1.f;
        dp[u] = j < Nk ? drow[j] * kp[u] : 0.f;                   // dP = dPd * keep
        dl += sv[u] * dp[u];
      }
      dl = wsum(dl);
#pragma unroll
      for (int u = 0; u < JMAX; ++u) {
        const int j = lane + 64 * u;
        if (j < NK4) {
          prow[j] = j < Nk ? sv[u] * kp[u] : 0.f;                 // Pd
          drow[j] = j < Nk ? sv[u] * (dp[u] - dl) * scale : 0.f;  // dS (scale folded)
        }
      }
    }
    __syncthreads();
    // dQ of this block, and this block's share of dK / dV into the thread's register tiles
    float* dQb = dQ + (size_t)(b * Nq + q0) * lddq + h * 64;
    for (int t = threadIdx.x; t < TM * 16; t += AT) {
      const int n0 = (t & 15) * 4, m0 = (t >> 4) * 4;
      float aq[4][4] = {};
      at_mix(Ds, pp, Ks, 64, NK4, m0, n0, aq);                    // dQ = dS K
      at_store_rows(dQb, lddq, m0, n0, nq, aq);
    }
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int t = threadIdx.x + u * AT;
      if (t < nkv) {
        const int n0 = (t & 15) * 4, m0 = (t >> 4) * 4;
        at_km(Ps, pp, Gs, 68, nq, m0, n0, av[u]);                 // dV += Pd^T dO
        at_km(Ds, pp, Qs, 68, nq, m0, n0, ak[u]);                 // dK += dS^T Q
      }
    }
  }
  float* dKb = dK + (size_t)b * Nk * lddk + h * 64;
  float* dVb = dV + (size_t)b * Nk * lddv + h * 64;
#pragma unroll
  for (int u = 0; u < KT; ++u) {
    const int t = threadIdx.x + u * AT;
    if (t < nkv) {
      const int n0 = (t & 15) * 4, m0 = (t >> 4) * 4;
      at_store_rows(dVb, lddv, m0, n0, Nk, av[u]);
      at_store_rows(dKb, lddk, m0, n0, Nk, ak[u]);
    }
  }
}

static size_t at_qblock_lds_bytes(int Nk, bool bwd) {
  const size_t k = (size_t)at_ru4(Nk) * 64;
  const int qb = bwd ? AT_QB : AT_QF;
  return 4 * ((bwd ? 2 : 1) * (size_t)qb * 68 + 2 * k + (bwd ? 2 : 1) * (size_t)qb * at_pitch(Nk));
}

static size_t at_lds_bytes(int Nq, int Nk, bool bwd) {
  const size_t q = (size_t)at_ru4(Nq) * 68, k = (size_t)at_ru4(Nk) * 64, p = (size_t)at_ru4(Nq) * at_pitch(Nk);
  return 4 * (bwd ? 2 * q + 2 * k + 2 * p : q + 2 * k + p);
}
static bool at_aligned(const void* p, int ld) { return ((uintptr_t)p & 15) == 0 && (ld & 3) == 0; }

int launch_attention_train_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, int B,
                               int Nq, int Nk, int H, float pdrop, unsigned long long seed, hipStream_t s) {
  PT_CHECK(Q && K && V && O && B > 0 && Nq > 0 && Nk > 0 && H > 0 && pdrop >= 0.f && pdrop < 1.f, "attention_train_fwd: bad argument");
  if (at_lds_bytes(Nq, Nk, false) <= 160 * 1024 && at_aligned(Q, ldq) && at_aligned(K, ldk) && at_aligned(V, ldv) &&
      at_aligned(O, ldo) && option(OPT_ATTN_TRAIN_ROWWISE) == 0) {
    PT_ENSURE_LDS(attn_fwd_tiled_kernel, 160 * 1024);
    PT_LAUNCH(attn_fwd_tiled_kernel, dim3(B * H), dim3(AT), at_lds_bytes(Nq, Nk, false), s, Q, ldq, K, ldk, V, ldv, O, ldo, Nq,
              Nk, H, 0.125f, pdrop, seed);
    return 0;
  }
  if (at_qblock_lds_bytes(Nk, false) <= 160 * 1024 && at_aligned(Q, ldq) && at_aligned(K, ldk) && at_aligned(V, ldv) &&
      at_aligned(O, ldo) && option(OPT_ATTN_TRAIN_ROWWISE) == 0) {
    PT_ENSURE_LDS(attn_fwd_qblock_kernel, 160 * 1024);
    PT_LAUNCH(attn_fwd_qblock_kernel, dim3(B * H, (Nq + AT_QF - 1) / AT_QF), dim3(AT), at_qblock_lds_bytes(Nk, false), s, Q, ldq, K,
              ldk, V, ldv, O, ldo, Nq, Nk, H, 0.125f, pdrop, seed);
    return 0;
  }
  PT_CHECK(Nk <= 4096, "attention_train_fwd: %d keys exceed the per-wave score buffer", Nk);
  PT_LAUNCH(attn_fwd_train_kernel, dim3((B * H * Nq + 3) / 4), dim3(256), (size_t)4 * Nk * sizeof(float), s, Q, ldq, K, ldk, V,
            ldv, O, ldo, B, Nq, Nk, H, 0.125f, pdrop, seed);
  return 0;
}
int launch_attention_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* dO, int ldo,
                         float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, float* lse, float* delta, int B,
                         int Nq, int Nk, int H, float pdrop, unsigned long long seed, hipStream_t s) {
  PT_CHECK(Q && K && V && dO && dQ && dK && dV && lse && delta && B > 0 && Nq > 0 && Nk > 0 && H > 0, "attention_bwd: bad argument");
  if (at_lds_bytes(Nq, Nk, true) <= 160 * 1024 && Nk <= 128 && at_aligned(Q, ldq) && at_aligned(K, ldk) && at_aligned(V, ldv) &&
      at_aligned(dO, ldo) && at_aligned(dQ, lddq) && at_aligned(dK, lddk) && at_aligned(dV, lddv) &&
      option(OPT_ATTN_TRAIN_ROWWISE) == 0) {
    PT_ENSURE_LDS(attn_bwd_tiled_kernel<2>, 160 * 1024);
    PT_LAUNCH(attn_bwd_tiled_kernel<2>, dim3(B * H), dim3(AT), at_lds_bytes(Nq, Nk, true), s, Q, ldq, K, ldk, V, ldv, dO, ldo, dQ,
              lddq, dK, lddk, dV, lddv, Nq, Nk, H, 0.125f, pdrop, seed);
    return 0;
  }
  if (at_qblock_lds_bytes(Nk, true) <= 160 * 1024 && Nk <= 256 && at_aligned(Q, ldq) && at_aligned(K, ldk) && at_aligned(V, ldv) &&
      at_aligned(dO, ldo) && at_aligned(dQ, lddq) && at_aligned(dK, lddk) && at_aligned(dV, lddv) &&
      option(OPT_ATTN_TRAIN_ROWWISE) == 0) {
    // Nk <= 256: four score columns per lane, at most 1,024 dK / dV tiles = two per thread
    PT_ENSURE_LDS((attn_bwd_qblock_kernel<4, 2>), 160 * 1024);
    PT_LAUNCH((attn_bwd_qblock_kernel<4, 2>), dim3(B * H), dim3(AT), at_qblock_lds_bytes(Nk, true), s, Q, ldq, K, ldk, V, ldv, dO, ldo,
              dQ, lddq, dK, lddk, dV, lddv, Nq, Nk, H, 0.125f, pdrop, seed);
    return 0;
  }
  PT_CHECK(Nk <= 2048, "attention_bwd: %d keys exceed the per-wave score buffer", Nk);
  const size_t smem = (size_t)4 * 2 * Nk * sizeof(float);
  PT_LAUNCH(attn_bwd_q_kernel, dim3((B * H * Nq + 3) / 4), dim3(256), smem, s, Q, ldq, K, ldk, V, ldv, dO, ldo, dQ, lddq, lse,
            delta, B, Nq, Nk, H, 0.125f, pdrop, seed);
  PT_LAUNCH(attn_bwd_kv_kernel, dim3((B * H * Nk + 3) / 4), dim3(256), 0, s, Q, ldq, K, ldk, V, ldv, dO, ldo, lse, delta, dK,
            lddk, dV, lddv, B, Nq, Nk, H, 0.125f, pdrop, seed);
  return 0;
}

// ---------------------------------------------------------------------------
// cross entropy over rows of V logits (F.cross_entropy, mean reduction): loss += (lse - logit[target]) / rows,
// dlogits = (softmax - onehot) / rows.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int V,
                                                 int rows, float* __restrict__ loss, float* __restrict__ dlogits) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* lg = logits + (size_t)row * V;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lg[c]);
  mx = wmax(mx);
  float l = 0.f;
  for (int c = lane; c < V; c += 64) l += expf(lg[c] - mx);
  l = wsum(l);
  const float ls = mx + logf(l);
  const int t = (int)target[row];
  const float inv = 1.f / (float)rows;
  for (int c = lane; c < V; c += 64)
    dlogits[(size_t)row * V + c] = (expf(lg[c] - ls) - (c == t ? 1.f : 0.f)) * inv;
  if (lane == 0) atomicAdd(loss, (ls - lg[t]) * inv);
}
int launch_ce(const float* logits, const int64_t* target, int V, int rows, float* loss, float* dlogits, hipStream_t s) {
  PT_CHECK(logits && target && loss && dlogits && V > 0 && rows > 0, "cross_entropy: bad argument");
  PT_LAUNCH(ce_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, target, V, rows, loss, dlogits);
  return 0;
}

// ---------------------------------------------------------------------------
// train-mode BatchNorm1d on channels-last rows (vqvae_modules.py:193-200 in train mode):
//   stats: per column mean and biased variance over the rows (fp64 accumulation), running-stat update;
//   apply: y = act((x - mean) * rstd * g + b (+ res));  bwd: dx, dgamma, dbeta (+ dres = gradient through act).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int ldx, int rows, int cols,
                                                       double* __restrict__ acc /* [2][cols] */, int rows_per_block) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  double s = 0.0, q = 0.0;
  if (c < cols)
    for (int r = r0 + sub; r < r1; r += 4) {
      const double v = x[(size_t)r * ldx + c];
      s += v; q += v * v;
    }
  __shared__ double red[2][4][64];
  red[0][sub][threadIdx.x & 63] = s;
  red[1][sub][threadIdx.x & 63] = q;
  __syncthreads();
  if (sub == 0 && c < cols) {
    atomicAdd(acc + c, red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x]);
    atomicAdd(acc + cols + c, red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x]);
  }
}
__global__ void bn_finish_kernel(const double* __restrict__ acc, int rows, int cols, float* __restrict__ mean_rstd /* [2][cols] */,
                                 float* __restrict__ run_mean, float* __restrict__ run_var, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const double m = acc[c] / rows;
  double var = acc[cols + c] / rows - m * m;
  var = var < 0 ? 0 : var;
  mean_rstd[c] = (float)m;
  mean_rstd[cols + c] = (float)(1.0 / sqrt(var + 1e-5));
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
    const double unb = rows > 1 ? var * rows / (rows - 1) : var;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
  }
}
// act: slope < 0 -> none; slope >= 0 -> leaky relu with that slope (0 = relu)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mean_rstd,
                                                       const float* __restrict__ g, const float* __restrict__ b,
                                                       const float* __restrict__ res, int ldr, float slope,
                                                       float* __restrict__ y, int ldy, int rows, int cols) {
  const size_t total = (size_t)rows * cols;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / cols), c = (int)(e - (size_t)r * cols);
    float v = (x[(size_t)r * ldx + c] - mean_rstd[c]) * mean_rstd[cols + c] * g[c] + b[c];
    if (res) v += res[(size_t)r * ldr + c];
    if (slope >= 0.f) v = v > 0.f ? v : v * slope;
    y[(size_t)r * ldy + c] = v;
  }
}
// dz = dy * act'(y) (gradient at the pre-activation sum BN(x) + res): written to dz (also the residual's gradient)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                      float slope, float* __restrict__ dz, int lddz, int rows, int cols) {
  const size_t total = (size_t)rows * cols;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / cols), c = (int)(e - (size_t)r * cols);
    const float d = dy[(size_t)r * lddy + c];
    dz[(size_t)r * lddz + c] = (slope < 0.f || y[(size_t)r * ldy + c] > 0.f) ? d : d * slope;
  }
}
// dx = g * rstd * (dz - mean(dz) - xhat * mean(dz * xhat)); sums = [sum dz | sum dz*xhat] per column (from colsum passes)
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ mean_rstd, const float* __restrict__ g,
                                                     const float* __restrict__ sums, float* __restrict__ dx, int lddx, int rows,
                                                     int cols) {
  const size_t total = (size_t)rows * cols;
  const float inv = 1.f / (float)rows;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / cols), c = (int)(e - (size_t)r * cols);
    const float rstd = mean_rstd[cols + c];
    const float xh = (x[(size_t)r * ldx + c] - mean_rstd[c]) * rstd;
    dx[(size_t)r * lddx + c] = g[c] * rstd * (dz[(size_t)r * lddz + c] - sums[c] * inv - xh * sums[cols + c] * inv);
  }
}
// dgamma needs sum dz * xhat: xhat is materialised on the fly: out[c] += sum_r dz * (x - mean) * rstd
__global__ __launch_bounds__(256) void bn_dgamma_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ mean_rstd, int rows, int cols,
                                                        float* __restrict__ sums /* [2][cols]: sum dz | sum dz*xhat */,
                                                        int rows_per_block) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  double s = 0.0, q = 0.0;
  if (c < cols) {
    const float m = mean_rstd[c], rs = mean_rstd[cols + c];
    for (int r = r0 + sub; r < r1; r += 4) {
      const double d = dz[(size_t)r * lddz + c];
      s += d;
      q += d * (double)((x[(size_t)r * ldx + c] - m) * rs);
    }
  }
  __shared__ double red[2][4][64];
  red[0][sub][threadIdx.x & 63] = s;
  red[1][sub][threadIdx.x & 63] = q;
  __syncthreads();
  if (sub == 0 && c < cols) {
    atomicAdd(sums + c, (float)(red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x]));
    atomicAdd(sums + cols + c, (float)(red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x]));
  }
}
static int ew_blocks(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
int launch_bn_train_fwd(const float* x, int ldx, int rows, int cols, const float* g, const float* b, const float* res, int ldr,
                        float slope, float* y, int ldy, float* mean_rstd, double* acc_ws, float* run_mean, float* run_var,
                        float momentum, hipStream_t s) {
  PT_CHECK(x && g && b && y && mean_rstd && acc_ws && rows > 0 && cols > 0, "bn_train_fwd: bad argument");
  PT_HIP(hipMemsetAsync(acc_ws, 0, sizeof(double) * 2 * cols, s));
  const int rpb = 256;
  PT_LAUNCH(bn_stats_kernel, dim3((cols + 63) / 64, (rows + rpb - 1) / rpb), dim3(256), 0, s, x, ldx, rows, cols, acc_ws, rpb);
  PT_LAUNCH(bn_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, acc_ws, rows, cols, mean_rstd, run_mean, run_var, momentum);
  PT_LAUNCH(bn_apply_kernel, dim3(ew_blocks((size_t)rows * cols)), dim3(256), 0, s, x, ldx, mean_rstd, g, b, res, ldr, slope, y,
            ldy, rows, cols);
  return 0;
}
// dy: gradient of the block output y = act(BN(x) + res).  Outputs: dz (gradient before the activation = residual gradient),
// dx, sums [2][cols] (= dbeta | dgamma, pre-zeroed by this call).
int launch_bn_train_bwd(const float* dy, int lddy, const float* y, int ldy, float slope, const float* x, int ldx,
                        const float* mean_rstd, const float* g, float* dz, int lddz, float* dx, int lddx, float* sums, int rows,
                        int cols, hipStream_t s) {
  PT_CHECK(dy && y && x && mean_rstd && g && dz && dx && sums && rows > 0 && cols > 0, "bn_train_bwd: bad argument");
  PT_LAUNCH(act_bwd_kernel, dim3(ew_blocks((size_t)rows * cols)), dim3(256), 0, s, dy, lddy, y, ldy, slope, dz, lddz, rows, cols);
  PT_HIP(hipMemsetAsync(sums, 0, sizeof(float) * 2 * cols, s));
  const int rpb = 256;
  PT_LAUNCH(bn_dgamma_kernel, dim3((cols + 63) / 64, (rows + rpb - 1) / rpb), dim3(256), 0, s, dz, lddz, x, ldx, mean_rstd, rows,
            cols, sums, rpb);
  PT_LAUNCH(bn_bwd_kernel, dim3(ew_blocks((size_t)rows * cols)), dim3(256), 0, s, dz, lddz, x, ldx, mean_rstd, g, sums, dx, lddx,
            rows, cols);
  return 0;
}
int launch_act_bwd(const float* dy, int lddy, const float* y, int ldy, float slope, float* dz, int lddz, int rows, int cols,
                   hipStream_t s) {
  PT_CHECK(dy && y && dz && rows > 0 && cols > 0, "act_bwd: bad argument");
  PT_LAUNCH(act_bwd_kernel, dim3(ew_blocks((size_t)rows * cols)), dim3(256), 0, s, dy, lddy, y, ldy, slope, dz, lddz, rows, cols);
  return 0;
}

// ---------------------------------------------------------------------------
// gate of GatedConv (gated_pixelcnn_1d.py:74-76): out = tanh(a) * sigmoid(b), hv = [a | b] (rows of 2*C)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ hv, int C, float* __restrict__ out, size_t total) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / C; const int c = (int)(e - r * C);
    const float a = hv[r * 2 * C + c], b = hv[r * 2 * C + C + c];
    out[e] = tanhf(a) * (1.f / (1.f + expf(-b)));
  }
}
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ hv, int C,
                                                       float* __restrict__ dhv, size_t total) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / C; const int c = (int)(e - r * C);
    const float a = hv[r * 2 * C + c], b = hv[r * 2 * C + C + c];
    const float t = tanhf(a), sg = 1.f / (1.f + expf(-b)), d = dout[e];
    dhv[r * 2 * C + c] = d * sg * (1.f - t * t);
    dhv[r * 2 * C + C + c] = d * t * sg * (1.f - sg);
  }
}
int launch_gate_fwd(const float* hv, int rows, int C, float* out, hipStream_t s) {
  PT_CHECK(hv && out && rows > 0 && C > 0, "gate_fwd: bad argument");
  PT_LAUNCH(gate_fwd_kernel, dim3(ew_blocks((size_t)rows * C)), dim3(256), 0, s, hv, C, out, (size_t)rows * C);
  return 0;
}
int launch_gate_bwd(const float* dout, const float* hv, int rows, int C, float* dhv, hipStream_t s) {
  PT_CHECK(dout && hv && dhv && rows > 0 && C > 0, "gate_bwd: bad argument");
  PT_LAUNCH(gate_bwd_kernel, dim3(ew_blocks((size_t)rows * C)), dim3(256), 0, s, dout, hv, C, dhv, (size_t)rows * C);
  return 0;
}

// ---------------------------------------------------------------------------
// token-embedding gradient: dE[id[row]] += dx[row]  (PTransformer.py:447; fp32 atomics)
// ---------------------------------------------------------------------------
// Workgroup (v, s) sums the rows of slice s whose id is v (ids scanned 64 at a time, matches walked by ballot), its 4
// waves meet in LDS and the row of dE[v] receives one atomic per slice: EMB_SLICES per entry instead of one per token
// (45,056 x 512 atomics on 129 rows took 2.3 ms).
constexpr int EMB_SLICES = 8;
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dx, const int* __restrict__ ids, int rows,
                                                        int vocab, float* __restrict__ dE) {
  __shared__ float red[4][DM];
  const int v = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per = (rows + EMB_SLICES * 4 - 1) / (EMB_SLICES * 4);
  const int r0 = (blockIdx.y * 4 + wv) * per, r1 = min(rows, r0 + per);
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  for (int base = r0; base < r1; base += 64) {
    const int row = base + lane;
    int id = row < r1 ? ids[row] : -1;
    if (row < r1) id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    unsigned long long m = __ballot(id == v);
    while (m) {
      const int k = __builtin_ctzll(m);
      m &= m - 1;
      const float* src = dx + (size_t)(base + k) * DM + lane * 4;
      a0 += *reinterpret_cast<const f32x4*>(src);
      a1 += *reinterpret_cast<const f32x4*>(src + 256);
    }
  }
  *reinterpret_cast<f32x4*>(&red[wv][lane * 4]) = a0;
  *reinterpret_cast<f32x4*>(&red[wv][256 + lane * 4]) = a1;
  __syncthreads();
  for (int c = threadIdx.x; c < DM; c += 256) {
    const float t = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    if (t != 0.f) atomicAdd(dE + (size_t)v * DM + c, t);
  }
}
int launch_embed_bwd(const float* dx, const int* ids, int rows, int vocab, float* dE, hipStream_t s) {
  PT_CHECK(dx && ids && dE && rows > 0 && vocab > 0, "embed_bwd: bad argument");
  PT_CHECK(((uintptr_t)dx & 15) == 0, "embed_bwd: dx must be 16-byte aligned");
  PT_LAUNCH(embed_bwd_kernel, dim3(vocab, EMB_SLICES), dim3(256), 0, s, dx, ids, rows, vocab, dE);
  return 0;
}

// ---------------------------------------------------------------------------
// dropout (train mode): mask from a counter-based hash of (seed, element); y = x * mask / (1-p).  The same call with
// the same seed replays the mask for the backward pass (no mask tensor is stored).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float p,
                                                      unsigned long long seed) {
  const float keep_scale = 1.f / (1.f - p);
  const unsigned thr = (unsigned)((double)p * 4294967296.0);
  const unsigned long long base = seed * 0x100000001B3ull;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    // four elements per lane and 16-byte accesses; the mask stays a function of (seed, element index), so any mix of
    // this path and the scalar tail replays the same mask
    const size_t n4 = n >> 2;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n4; v += (size_t)gridDim.x * 256) {
      const f32x4 in = *reinterpret_cast<const f32x4*>(x + 4 * v);
      f32x4 out;
#pragma unroll
      for (int i = 0; i < 4; ++i) out[i] = mix32(base + 4 * v + i) >= thr ? in[i] * keep_scale : 0.f;
      *reinterpret_cast<f32x4*>(y + 4 * v) = out;
    }
    for (size_t e = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
      y[e] = mix32(base + e) >= thr ? x[e] * keep_scale : 0.f;
    return;
  }
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
    y[e] = mix32(base + e) >= thr ? x[e] * keep_scale : 0.f;
}
int launch_dropout(const float* x, float* y, size_t n, float p, unsigned long long seed, hipStream_t s) {
  PT_CHECK(x && y && n > 0 && p >= 0.f && p < 1.f, "dropout: bad argument");
  PT_LAUNCH(dropout_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, n, p, seed);
  return 0;
}

// ---------------------------------------------------------------------------
// optimizer: squared-norm accumulation, gradient scaling, AdamW (torch.optim.AdamW semantics, amsgrad off)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ out) {
  float s = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) s += g[e] * g[e];
  s = wsum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
int launch_sumsq(const float* g, size_t n, float* out, hipStream_t s) {
  PT_CHECK(g && out && n > 0, "sumsq: bad argument");
  size_t b = (n + 1023) / 1024;
  PT_LAUNCH(sumsq_kernel, dim3((unsigned)(b > 1024 ? 1024 : b)), dim3(256), 0, s, g, n, out);
  return 0;
}
// p, m, v updated in place; the gradient is multiplied by *gscale (device scalar: the clip coefficient) first
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, const float* __restrict__ gscale, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt) {
  const float gs = gscale ? *gscale : 1.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const float gr = g[e] * gs;
    float pv = p[e] * (1.f - lr * wd);
    const float mv = beta1 * m[e] + (1.f - beta1) * gr;
    const float vv = beta2 * v[e] + (1.f - beta2) * gr * gr;
    m[e] = mv; v[e] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pv -= (lr / bc1) * (mv / denom);
    p[e] = pv;
  }
}
int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, const float* gscale, float lr, float beta1, float beta2,
                 float eps, float wd, int step, hipStream_t s) {
  PT_CHECK(p && g && m && v && n > 0 && step >= 1, "adamw: bad argument");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  PT_LAUNCH(adamw_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, p, g, m, v, n, gscale, lr, beta1, beta2, eps, wd, bc1, sqrtf(bc2));
  return 0;
}
// clip coefficient of torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (sqrt(sumsq) + 1e-6)); norm_out = sqrt(sumsq)
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef, float* __restrict__ norm_out) {
  const float nrm = sqrtf(*sumsq);
  const float c = max_norm / (nrm + 1e-6f);
  *coef = c < 1.f ? c : 1.f;
  if (norm_out) *norm_out = nrm;
}
int launch_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, hipStream_t s) {
  PT_CHECK(sumsq && coef, "clip_coef: bad argument");
  PT_LAUNCH(clip_coef_kernel, dim3(1), dim3(1), 0, s, sumsq, max_norm, coef, norm_out);
  return 0;
}

// out = a + b (row-major, equal shapes) and out *= scalar: the two elementwise ops the autograd glue needs
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) out[e] = a[e] + b[e];
}
int launch_add(const float* a, const float* b, float* out, size_t n, hipStream_t s) {
  PT_CHECK(a && b && out && n > 0, "add: bad argument");
  PT_LAUNCH(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, a, b, out, n);
  return 0;
}

// ---------------------------------------------------------------------------
// PQ-VAE training pieces (vq_teacher.py:135-232, vqvae_modules.py:495-544).
// ---------------------------------------------------------------------------
// EMA codebook statistics: counts[g][k] = #rows assigned, dw[g][k][:] = sum of their latent chunks (the reference's
// one_hot^T . flat_x, vqvae_modules.py:515-520).  One wave per (row, group).
// shared = 1 (share_code, 496-505): every group's chunks count towards the one code book, statistics [K] and [K][D].
__global__ __launch_bounds__(256) void pq_ema_accum_kernel(const float* __restrict__ z, int ldz, const int64_t* __restrict__ idx,
                                                           int rows, int G, int D, int K, float* __restrict__ counts,
                                                           float* __restrict__ dw, int shared) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (size_t w = (size_t)blockIdx.x * 4 + wave; w < (size_t)rows * G; w += (size_t)gridDim.x * 4) {
    const int row = (int)(w / G), g = (int)(w % G);
    const int64_t k = idx[(size_t)row * G + g];
    if (k < 0 || k >= K) continue;
    const int gs = shared ? 0 : g;
    float* dst = dw + ((size_t)gs * K + k) * D;
    const float* src = z + (size_t)row * ldz + g * D;
    for (int d = lane; d < D; d += 64) atomicAdd(dst + d, src[d]);
    if (lane == 0) atomicAdd(counts + (size_t)gs * K + k, 1.f);
  }
}
// hidden -= (hidden - value)(1 - decay); average = hidden / (1 - decay^counter); Laplace smoothing of the cluster sizes;
// codes whose smoothed size < 1 are re-seeded from the first K latent rows (`_tile`, 571-580) and their EMA state reset
// (529-534).  A batch with fewer chunks than codes is tiled with noise as `_tile` does (573-577): seed row k is chunk
// k mod nb plus N(0,1) * 0.01 / sqrt(D), the N(0,1) draw taken from `noise` [groups][K][D] when given (the parity mode:
// the first K rows of each torch.randn_like of the reference) or from a counter-based generator keyed by (seed, group,
// code, column).  One workgroup per group, one thread per code.
__device__ __forceinline__ float tile_normal(unsigned long long seed, unsigned g, unsigned k, unsigned d) {
  // two 24-bit uniforms from a 64-bit mix of the counter, Box-Muller
  unsigned long long x = seed ^ (0x9E3779B97F4A7C15ull * (((unsigned long long)g << 40) ^ ((unsigned long long)k << 20) ^ d) + 0xD1B54A32D192ED03ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  const float u1 = ((float)((x >> 40) & 0xFFFFFF) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
  const float u2 = (float)((x >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}
__global__ __launch_bounds__(1024) void pq_ema_finish_kernel(const float* __restrict__ counts, const float* __restrict__ dw,
                                                             const float* __restrict__ z, int ldz, int D, int K,
                                                             float* __restrict__ cs_hidden, float* __restrict__ dw_hidden,
                                                             float* __restrict__ emb, float one_minus_decay, float eps,
                                                             float corr, int reseed, int rows, int nb, const float* __restrict__ noise,
                                                             float tile_std, unsigned long long tile_seed) {
  // reseed = 1: ProductQuantization (dead codes re-seeded and their EMA state reset, 525-534); 0: VectorQuantizerEMA
  // (323-334, no re-seeding); 2: ProductQuantization with share_code (509-512): dead codes of the one code book are
  // re-seeded from the first K chunks of the group-major flat_x of line 498 (chunk k = group k / rows, row k % rows), the
  // EMA state is kept.
  // A thread walks codes k, k + blockDim.x, ... (code books larger than one workgroup).
  __shared__ float red[16];
  const int g = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float part = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float h = cs_hidden[(size_t)g * K + k];
    h -= (h - counts[(size_t)g * K + k]) * one_minus_decay;
    part += h / corr;
  }
  part = wsum(part);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  float n = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) n += red[w];
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float h = cs_hidden[(size_t)g * K + k];
    h -= (h - counts[(size_t)g * K + k]) * one_minus_decay;
    const float avg = h / corr;
    const float sm = (avg + eps) / (n + (float)K * eps) * n;
    const bool used = !reseed || sm >= 1.0f;
    // nb = chunks `_tile` sees (rows, or rows * G when the code book is shared); seed chunk of code k: k mod nb
    const bool tiled = reseed && nb < K;
    const int kc = tiled ? k % nb : k;
    const int seed_g = reseed == 2 ? kc / rows : g, seed_row = reseed == 2 ? kc - seed_g * rows : kc;
    for (int d = 0; d < D; ++d) {
      const size_t o = ((size_t)g * K + k) * D + d;
      float hd = dw_hidden[o];
      hd -= (hd - dw[o]) * one_minus_decay;
      const float nrm = (hd / corr) / sm;
      float rnd = reseed ? z[(size_t)seed_row * ldz + seed_g * D + d] : 0.f;
      if (tiled) {
        // out + randn_like(out) * std (577): a rounded product, then a rounded sum (no fma), as torch computes it
        const float nz = noise ? noise[o] : tile_normal(tile_seed, (unsigned)g, (unsigned)k, (unsigned)d);
        // (hipcc contracts even __fmul_rn + __fadd_rn into one fma: the product goes through an opaque register first)
        float prod = nz * tile_std;
        asm volatile("" : "+v"(prod));
        rnd = rnd + prod;
      }
      dw_hidden[o] = (used || reseed == 2) ? hd : rnd;
      emb[o] = used ? nrm : rnd;
    }
    cs_hidden[(size_t)g * K + k] = (used || reseed == 2) ? h : 1.0f;
  }
}
int launch_pq_ema(const float* z, int ldz, const int64_t* idx, int rows, int G, int D, int K, float* counts, float* dw,
                  float* cs_hidden, float* dw_hidden, float* emb, float decay, float eps, int counter, hipStream_t s, int reseed,
                  const float* tile_noise, unsigned long long tile_seed) {
  PT_CHECK(z && idx && counts && dw && cs_hidden && dw_hidden && emb && G > 0 && D > 0 && K > 0 && rows > 0, "pq_ema: bad argument");
  const long long nb = reseed == 2 ? (long long)rows * G : (long long)rows;    // chunks `_tile` sees
  PT_CHECK(counter >= 1 && decay > 0.f && decay < 1.f, "pq_ema: bad decay / counter");
  const int shared = reseed == 2, Gs = shared ? 1 : G;            // statistics and state of ONE code book when shared
  PT_HIP(hipMemsetAsync(counts, 0, sizeof(float) * Gs * K, s));
  PT_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Gs * K * D, s));
  const size_t waves = (size_t)rows * G;
  PT_LAUNCH(pq_ema_accum_kernel, dim3((unsigned)std::min<size_t>((waves + 3) / 4, 65535)), dim3(256), 0, s, z, ldz, idx, rows, G,
            D, K, counts, dw, shared);
  const float omd = (float)(1.0 - (double)decay);
  const float corr = (float)(1.0 - std::pow((double)decay, (double)counter));
  const float tile_std = (float)(0.01 / std::sqrt((double)D));                  // vqvae_modules.py:575
  PT_LAUNCH(pq_ema_finish_kernel, dim3(Gs), dim3(std::min((K + 63) / 64 * 64, 1024)), 0, s, counts, dw, z, ldz, D, K, cs_hidden,
            dw_hidden, emb, omd, eps, corr, reseed, rows, (int)std::min<long long>(nb, 1 << 30), tile_noise, tile_std, tile_seed);
  return 0;
}

// Reconstruction losses of vq_teacher.get_loss (184-232): per-part smooth-L1 (beta) means over column ranges, weighted,
// plus the velocity L1 mean over the first vel_cols columns; writes d(total)/d(pred).
// sums[p] (p < nparts) = part means, sums[nparts] = velocity mean, sums[nparts+1] = weighted total.
struct VqLossArgs {
  const float* pred; int ldp;
  const float* gt; int ldg;
  float* dpred; int lddp;
  int B, T, C, nparts, vel_cols;
  int bounds[5];
  float w[4];        // weight of each part in the total
  float inv_cnt[4];  // 1 / (B*T*cols of the part)
  float inv_vel;     // 1 / (B*(T-1)*vel_cols)
  float beta;
  float* sums;
  const float* nmask;  // per-row weight (1 - mask) of the refiner's masked L1 (refiner.py:337-349), or NULL
  int l1;              // 0: smooth-L1(beta) (vq_teacher), 1: L1 * nmask (refiner)
};
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) - (v < 0.f); }
__global__ __launch_bounds__(256) void vq_loss_kernel(const VqLossArgs a) {
  __shared__ float red[4][6];
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const size_t total = (size_t)a.B * a.T * a.C;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % a.C);
    const size_t row = e / a.C;
    const int t = (int)(row % a.T);
    const float p = a.pred[row * a.ldp + c], g = a.gt[row * a.ldg + c];
    const float d = p - g, ad = fabsf(d);
    int part = 0;
    while (part + 1 < a.nparts && c >= a.bounds[part + 1]) ++part;
    float l, grad;
    if (a.l1) {
      const float wr = a.nmask ? a.nmask[row] : 1.f;
      l = ad * wr;
      grad = sgn(d) * wr * a.inv_cnt[part] * a.w[part];
    } else {
      l = ad < a.beta ? 0.5f * d * d / a.beta : ad - 0.5f * a.beta;
      grad = (ad < a.beta ? d / a.beta : sgn(d)) * a.inv_cnt[part] * a.w[part];
    }
    acc[part] += l * a.inv_cnt[part];
    if (c < a.vel_cols) {
      if (t + 1 < a.T) {
        const float vd = (a.pred[(row + 1) * a.ldp + c] - p) - (a.gt[(row + 1) * a.ldg + c] - g);
        acc[4] += fabsf(vd) * a.inv_vel;
        grad -= sgn(vd) * a.inv_vel;
      }
      if (t > 0) {
        const float vd = (p - a.pred[(row - 1) * a.ldp + c]) - (g - a.gt[(row - 1) * a.ldg + c]);
        grad += sgn(vd) * a.inv_vel;
      }
    }
    a.dpred[row * a.lddp + c] = grad;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float v = wsum(acc[i]);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    const int i = threadIdx.x;
    if (i < a.nparts) {
      atomicAdd(a.sums + i, v);
      atomicAdd(a.sums + a.nparts + 1, v * a.w[i]);
    } else if (i == 4) {
      atomicAdd(a.sums + a.nparts, v);
      atomicAdd(a.sums + a.nparts + 1, v);
    }
  }
}
int launch_vq_losses(const float* pred, int ldp, const float* gt, int ldg, int B, int T, int nparts, const int* bounds,
                     const float* weights, int vel_cols, float beta, float* sums, float* dpred, int lddp, hipStream_t s,
                     const float* nmask, float t_nmask, int l1) {
  PT_CHECK(pred && gt && sums && dpred && bounds && weights && B > 0 && T > 1, "vq_losses: bad argument");
  PT_CHECK(!l1 || t_nmask > 0.f, "masked_l1_losses: no unmasked frame");
  PT_CHECK(nparts >= 1 && nparts <= 4, "vq_losses: 1..4 parts, got %d", nparts);
  VqLossArgs a{};
  a.pred = pred; a.ldp = ldp; a.gt = gt; a.ldg = ldg; a.dpred = dpred; a.lddp = lddp;
  a.B = B; a.T = T; a.nparts = nparts; a.vel_cols = vel_cols; a.beta = beta; a.sums = sums;
  a.nmask = nmask; a.l1 = l1;
  for (int i = 0; i <= nparts; ++i) a.bounds[i] = bounds[i];
  a.C = bounds[nparts];
  PT_CHECK(bounds[0] == 0 && a.C <= ldp && a.C <= ldg && a.C <= lddp && vel_cols >= 0 && vel_cols <= a.C, "vq_losses: bad column ranges");
  for (int i = 0; i < nparts; ++i) {
    PT_CHECK(bounds[i + 1] > bounds[i], "vq_losses: empty part %d", i);
    a.w[i] = weights[i];
    a.inv_cnt[i] = (float)(1.0 / ((l1 ? (double)t_nmask : (double)B * T) * (bounds[i + 1] - bounds[i])));
  }
  a.inv_vel = vel_cols > 0 ? (float)(1.0 / ((double)B * (T - 1) * vel_cols)) : 0.f;
  PT_HIP(hipMemsetAsync(sums, 0, sizeof(float) * (nparts + 2), s));
  PT_LAUNCH(vq_loss_kernel, dim3(ew_blocks((size_t)B * T * a.C)), dim3(256), 0, s, a);
  return 0;
}

// commitment loss pieces: *out += sum (a-b)^2 ;  dz = dq + (*g) * coef * (z - q)
__global__ __launch_bounds__(256) void sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                     float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const float d = a[e] - b[e];
    acc += d * d;
  }
  acc = wsum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void commit_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ z,
                                                         const float* __restrict__ q, const float* __restrict__ g, float coef,
                                                         size_t n, float* __restrict__ dz) {
  const float c = (g ? *g : 1.f) * coef;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
    dz[e] = (dq ? dq[e] : 0.f) + c * (z[e] - q[e]);
}
int launch_sqdiff(const float* a, const float* b, size_t n, float* out, hipStream_t s) {
  PT_CHECK(a && b && out && n > 0, "sqdiff: bad argument");
  PT_LAUNCH(sqdiff_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, a, b, n, out);
  return 0;
}
int launch_commit_bwd(const float* dq, const float* z, const float* q, const float* g, float coef, size_t n, float* dz,
                      hipStream_t s) {
  PT_CHECK(z && q && dz && n > 0, "commit_bwd: bad argument");
  PT_LAUNCH(commit_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dq, z, q, g, coef, n, dz);
  return 0;
}

}  // namespace pt
