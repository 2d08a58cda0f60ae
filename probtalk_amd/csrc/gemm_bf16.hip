// tap-GEMM on the gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16): bf16 operands, fp32
// accumulate, fp32 epilogue (bias, fp64 PE, residual, LeakyReLU/ReLU) and fp32 and/or bf16 output.
// Same contract as gemm.hip (taps = row-shifted reads of a channels-last activation); used for the
// predictor / refiner transformer in bf16 mode (BASELINE configs[2]: bf16, 1e-2 tolerance).
//
// Tiling: 256 threads = 4 waves (2x2), wave tile MI x NI of 32x32, BK = 64 bf16 (128 B rows, padded
// to 144 B in LDS so the 16-lane groups of ds_read_b128 hit 16 different bank quads), double-buffered
// LDS with register prefetch, one barrier per stage.  One b128 fragment read feeds one MFMA per
// operand (8 bf16 = lane's k-slice of a 32x32x16 step).
#include <vector>

#include "pt_common.h"

namespace pt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BKB = 64;          // bf16 elements per stage
constexpr int LDS_STB = 72;      // bf16 elements per LDS row (144 B)

template <int MI, int NI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgsB p, const int tiles_n, const int nwg) {
  constexpr int BM = 64 * MI, BN = 64 * NI;
  constexpr int AI = 2 * MI, BI = 2 * NI;  // 16-byte loads per thread per stage
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_b[];
  constexpr int STAGE = (BM + BN) * LDS_STB;

  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c8 = tid & 7, r0 = tid >> 3;

  int a_l[AI], a_rb[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      const int b = m / p.Lgrid;
      a_l[i] = (m - b * p.Lgrid) * p.in_stride;
      a_rb[i] = b * p.Lin;
    } else {
      a_l[i] = -(1 << 28);
      a_rb[i] = 0;
    }
  }
  const int kch = p.K / BKB;
  const int nk = p.ntaps * kch;

  u32x4 ra[AI], rb[BI];
  int tap = 0, kc = 0;
  auto gload = [&]() {
    const int d = p.tap_d[tap];
    const unsigned short* Wt = p.W + (size_t)p.tap_w[tap] * p.Npad * p.K;
    const int k = kc * BKB + c8 * 8;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int l = a_l[i] + d;
      if ((unsigned)l < (unsigned)p.Lin)
        ra[i] = *reinterpret_cast<const u32x4*>(p.A + (size_t)(a_rb[i] + l) * p.lda + k);
      else
        ra[i] = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < BI; ++i)
      rb[i] = *reinterpret_cast<const u32x4*>(Wt + (size_t)(n0 + r0 + 32 * i) * p.K + k);
    if (++kc == kch) { kc = 0; ++tap; }
  };
  auto lstore = [&](int buf) {
    unsigned short* As = smem_b + buf * STAGE;
    unsigned short* Bs = As + BM * LDS_STB;
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<u32x4*>(As + (r0 + 32 * i) * LDS_STB + c8 * 8) = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<u32x4*>(Bs + (r0 + 32 * i) * LDS_STB + c8 * 8) = rb[i];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  gload();
  lstore(0);
  __syncthreads();

  const int frag_off = (lane & 31) * LDS_STB + 8 * (lane >> 5);
  for (int it = 0; it < nk; ++it) {
    const bool more = it + 1 < nk;
    if (more) gload();
    const unsigned short* As = smem_b + (it & 1) * STAGE + (wm * 32 * MI) * LDS_STB + frag_off;
    const unsigned short* Bs = smem_b + (it & 1) * STAGE + (BM + wn * 32 * NI) * LDS_STB + frag_off;
#pragma unroll
    for (int s = 0; s < BKB / 16; ++s) {
      bf16x8 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(As + mi * 32 * LDS_STB + s * 16);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(Bs + ni * 32 * LDS_STB + s * 16);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    if (more) lstore((it + 1) & 1);
    __syncthreads();
  }

  const bool plain_rows = (p.out_stride == 1 && p.out_off == 0 && p.Lgrid == p.Lout);
  const int lcol = lane & 31, lrow = 4 * (lane >> 5);
  const bool gated = (p.act == ACT_GATED);
  const int n_out = gated ? p.N / 2 : p.N;
  auto put = [&](int orow, int col, float v) {
    if (p.C) p.C[(size_t)orow * p.ldc + col] = v;
    if (p.Cb) reinterpret_cast<__bf16*>(p.Cb)[(size_t)orow * p.ldcb + col] = (__bf16)v;
  };
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 * MI + mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
      if (m >= p.M) continue;
      int orow = m;
      if (!plain_rows) {
        const int b = m / p.Lgrid;
        orow = b * p.Lout + (m - b * p.Lgrid) * p.out_stride + p.out_off;
      }
      if (gated) {
        // weights are interleaved in 32-channel blocks [a | b] (runtime.hip pack_conv): tanh(a) * sigmoid(b)
        if constexpr (NI % 2 == 0) {
#pragma unroll
          for (int pr = 0; pr < NI / 2; ++pr) {
            const int nperm = n0 + wn * 32 * NI + pr * 64 + lcol;
            const int col = (n0 + wn * 32 * NI + pr * 64) / 2 + lcol;
            if (col < n_out) {
              float va = acc[mi][2 * pr][r], vb = acc[mi][2 * pr + 1][r];
              if (p.bias) { va += p.bias[nperm]; vb += p.bias[nperm + 32]; }
              put(orow, col, tanhf(va) * (1.f / (1.f + expf(-vb))));
            } else if (col < p.zero_to) {
              put(orow, col, 0.f);
            }
          }
        }
      } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int col = n0 + wn * 32 * NI + ni * 32 + lcol;
          if (col < n_out) {
            float v = acc[mi][ni][r];
            if (p.bias) v += p.bias[col];
            if (p.pe) v = (float)((double)v + p.pe[(size_t)(orow % p.pe_period) * p.N + col]);
            if (p.Res) v += p.Res[(size_t)orow * p.ldr + col];
            if (p.ResB) v += (float)reinterpret_cast<const __bf16*>(p.ResB)[(size_t)orow * p.ldr + col];
            if (p.act == ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
            put(orow, col, v);
          } else if (col < p.zero_to) {
            put(orow, col, 0.f);
          }
        }
      }
    }
  }
}

void prof_record_begin(hipStream_t s, double flops, int cfg, void** tok);
void prof_record_end(hipStream_t s, void* tok);
bool prof_is_on();

template <int MI, int NI>
static int launch_cfg_b(const GemmArgsB& a, hipStream_t s) {
  constexpr int BM = 64 * MI, BN = 64 * NI;
  constexpr size_t smem = 2 * (size_t)(BM + BN) * LDS_STB * sizeof(unsigned short);
  PT_ENSURE_LDS((gemm_bf16_kernel<MI, NI>), smem);
  const int tiles_m = (a.M + BM - 1) / BM;
  const int tiles_n = (a.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  void* tok = nullptr;
  if (prof_is_on()) prof_record_begin(s, a.flops > 0 ? a.flops : 2.0 * a.M * a.N * a.K * a.ntaps,
                                      3 + ((MI == 2 && NI == 2) ? 0 : (NI == 2 ? 1 : 2)), &tok);
  PT_LAUNCH((gemm_bf16_kernel<MI, NI>), dim3(nwg), dim3(256), smem, s, a, tiles_n, nwg);
  if (tok) prof_record_end(s, tok);
  return 0;
}

int launch_gemm_bf16(const GemmArgsB& a, hipStream_t s) {
  PT_CHECK(a.M > 0 && a.N > 0, "gemm_bf16: empty problem M=%d N=%d", a.M, a.N);
  PT_CHECK(a.K % BKB == 0 && a.K > 0, "gemm_bf16: K=%d must be a positive multiple of %d", a.K, BKB);
  PT_CHECK(a.lda >= a.K && a.lda % 8 == 0, "gemm_bf16: lda=%d must be >= K=%d and a multiple of 8", a.lda, a.K);
  PT_CHECK(a.Npad % 128 == 0 && a.Npad >= a.N, "gemm_bf16: Npad=%d must be a multiple of 128 and >= N=%d", a.Npad, a.N);
  PT_CHECK(a.ntaps >= 1 && a.ntaps <= MAX_TAPS, "gemm_bf16: ntaps=%d out of range", a.ntaps);
  PT_CHECK(a.Lgrid > 0 && a.M % a.Lgrid == 0, "gemm_bf16: M=%d is not a multiple of Lgrid=%d", a.M, a.Lgrid);
  PT_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm_bf16: A and W must be 16-byte aligned");
  PT_CHECK(a.C || a.Cb, "gemm_bf16: no output");
  const int n_out = a.act == ACT_GATED ? a.N / 2 : a.N;
  PT_CHECK((!a.C || (a.ldc >= n_out && a.zero_to <= a.ldc)) && (!a.Cb || (a.ldcb >= n_out && a.zero_to <= a.ldcb)),
           "gemm_bf16: output leading dim too small for N=%d / zero_to=%d", n_out, a.zero_to);
  PT_CHECK(!(a.Res && a.ResB), "gemm_bf16: pass the residual as fp32 or bf16, not both");
  if (a.act == ACT_GATED) {
    PT_CHECK(a.N % 64 == 0, "gemm_bf16: gated N=%d must be a multiple of 64", a.N);
    PT_CHECK(!a.pe && !a.Res && !a.ResB, "gemm_bf16: gated epilogue takes no pe/residual");
  }
  if (option(OPT_CONV8) && conv8_takes(a)) return launch_conv8(a, s);
  auto tiles = [&](int bm, int bn) { return ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  if (tiles(128, 128) >= 512) return launch_cfg_b<2, 2>(a, s);
  if (tiles(64, 128) >= 384 || a.act == ACT_GATED) return launch_cfg_b<1, 2>(a, s);
  return launch_cfg_b<1, 1>(a, s);
}

}  // namespace pt
