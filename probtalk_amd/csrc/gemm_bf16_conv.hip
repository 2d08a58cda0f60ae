// Phase-pipelined 256 x 256 bf16 tap-GEMM for the conv stacks of the bf16 mode (PQ-VAE decoder: Res_CNR_Stack k3 convs,
// ConvTranspose up-blocks, 1x1 projections, nets/spg/vqvae_modules.py:193-245, nets/inpainting/vqvae_1d_sc.py:259-274;
// audio ConditionEncoder, vqvae_1d_sc.py:86-91): C[orow(m), n] = act(sum_t sum_k A[src(m, t), k] W[t][n][k] + bias (+ res)).
//
// The schedule is gemm_bf16_8ph.hip's (8 waves 2 x 4, four half-tile LDS-DMA images per 64-deep K-tile, four MFMA phases
// per K-tile, waves 4-7 one barrier behind, persistent workgroups, K-tile stream across tiles); what is new is the K axis:
// K-tile kt of a tile is (tap = kt / kch, channel block kt % kch), and the A half-tile images of tap t are the SAME
// activation rows shifted by tap_d[t] frames.  A frame outside its clip must read zeros (conv padding): LDS-DMA cannot
// mask a lane to zero, so such a lane's source address is redirected to a 128-byte zero block in device memory (per-lane
// 64-bit addresses instead of base + 32-bit offset).  The activation is read from L2 after the first tap: the three (k3)
// to four (k4 s2) taps of a 256-row tile are issued within ~10 us of each other.
// Replaces the 4-wave register-staged gemm_bf16_kernel for these shapes (measured there: 280 TFLOP/s, 2-byte stores).
#include <cstdlib>

#include <mutex>

#include "pt_common.h"

namespace pt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

void prof_record_begin(hipStream_t s, double flops, int cfg, void** tok);
void prof_record_end(hipStream_t s, void* tok);
bool prof_is_on();

constexpr int C8_HT = 128 * 128;          // bytes of a half-tile image (128 rows x 64 bf16)
constexpr int C8_STAGE = 4 * C8_HT;
constexpr int C8_RING = 2 * C8_STAGE;
constexpr int C8_BIAS_COLS = 4096;        // bias row kept in LDS (fp32), see gemm_bf16_8ph.hip
constexpr int C8_SMEM = C8_RING + C8_BIAS_COLS * 4;

__device__ __forceinline__ unsigned c8_pack(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

__global__ __launch_bounds__(512) void conv8_kernel(const GemmArgsB p, const unsigned char* __restrict__ zero,
                                                    const int tiles_n, const int ntiles) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int G = gridDim.x;
  int tile0;
  {
    const int q = G >> 3, rr = G & 7, x = blockIdx.x & 7;       // bijective XCD-aware order (see gemm_bf16_big.hip)
    tile0 = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (blockIdx.x >> 3);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, q4 = lane >> 4;
  const int kch = p.K >> 6;                 // 64-deep channel blocks per tap
  const int nt = p.ntaps * kch;             // K-tiles per output tile

  // ---- per-half-tile DMA cursors (wave-uniform) and per-lane sources
  int c_tile[4], c_kc[4], c_tap[4], c_buf[4];
  int a_l[2][2], a_row[2][2];               // A images: clip-local source frame of tap offset 0 (or far negative), global source row
  unsigned soffW[2][2];                     // W images: byte offset of the lane's row chunk inside one tap slab
  // chunk swizzle of the lane's image row (the same for both instructions of a half-tile: rows differ by 64)
  const int gch = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
  auto set_src = [&](int h, int tl) {
    const int tm = tl / tiles_n, tn = tl - tm * tiles_n;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (wave + 8 * i) * 8 + (lane >> 3);              // local row of the half-tile image
      if (h < 2) {
        const int m = tm * 256 + (lr >> 6) * 128 + h * 64 + (lr & 63);
        if (m < p.M) {
          const int b = m / p.Lgrid;
          const int l0 = (m - b * p.Lgrid) * p.in_stride;
          a_l[h][i] = l0;
          a_row[h][i] = b * p.Lin + l0;
        } else {
          a_l[h][i] = -(1 << 28);
          a_row[h][i] = 0;
        }
      } else {
        int n = tn * 256 + (lr >> 5) * 64 + (h - 2) * 32 + (lr & 31);
        n = n < p.Npad ? n : p.Npad - 1;
        soffW[h - 2][i] = ((unsigned)n * (unsigned)p.K + gch * 8) * 2u;
      }
    }
  };
  bool draining = false;
  auto issue = [&](int h) {
    if (c_tile[h] >= ntiles) {
      draining = true;
      return;
    }
    const int tap = c_tap[h], kc = c_kc[h];
    unsigned char* dst = smem + c_buf[h] * C8_STAGE + h * C8_HT + wave * 1024;
    if (h < 2) {
      const int d = p.tap_d[tap];
      const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.A) + kc * 128 + gch * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = (unsigned)(a_l[h][i] + d) < (unsigned)p.Lin;
        const unsigned char* src = ok ? abase + (size_t)(unsigned)(a_row[h][i] + d) * (size_t)(p.lda * 2) : zero + gch * 16;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(dst + i * 8 * 1024), 16, 0, 0);
      }
    } else {
      const unsigned char* wbase =
          reinterpret_cast<const unsigned char*>(p.W) + ((size_t)p.tap_w[tap] * p.Npad * p.K + (size_t)kc * 64) * 2;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(wbase + soffW[h - 2][0]), (lds_ptr_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(wbase + soffW[h - 2][1]), (lds_ptr_t)(dst + 8 * 1024), 16, 0, 0);
    }
    c_buf[h] ^= 1;
    if (++c_kc[h] == kch) {
      c_kc[h] = 0;
      if (++c_tap[h] == p.ntaps) {
        c_tap[h] = 0;
        c_tile[h] += G;
        if (c_tile[h] < ntiles) set_src(h, c_tile[h]);
      }
    }
  };
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    c_tile[h] = tile0;
    c_kc[h] = 0;
    c_tap[h] = 0;
    c_buf[h] = 0;
    set_src(h, tile0);
  }
  // prologue: K-tile 0 whole, then h0 / h2 of K-tile 1 (the loop issues h3 / h1 of t+1 and h0 / h2 of t+2)
  issue(0); issue(2); issue(3); issue(1); issue(0); issue(2);
  float* bias_s = reinterpret_cast<float*>(smem + C8_RING);
  for (int n = tid * 4; n < tiles_n * 256; n += 2048) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + n);      // N % 8 == 0 (conv8_takes)
    *reinterpret_cast<f32x4*>(bias_s + n) = bv;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();                    // waves 4-7 run one barrier behind

  // fragment read offsets inside a half-tile image: row r, chunk (4 s + q4) ^ ((r >> 1) & 7)
  int aoff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) aoff[s] = r * 128 + (((4 * s + q4) ^ ((r >> 1) & 7)) << 4);
  const int a_row0 = wr * 64 * 128, b_row0 = wc * 32 * 128;

  f32x4 acc[8][4];
  bf16x8 af[4][2], bfr[2][2][2];
  int cbuf = 0;

  for (int tile = tile0; tile < ntiles; tile += G) {
    {
      int iq = q4;
      asm volatile("" : "+v"(iq));
      const int tn0 = tile - (tile / tiles_n) * tiles_n;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = tn0 * 256 + wc * 64 + j * 16 + 4 * iq;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + n);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][j] = bv;
      }
    }

    for (int t = 0; t < nt; ++t, cbuf ^= 1) {
      const unsigned char* sb = smem + cbuf * C8_STAGE;
      auto readA = [&](int mh) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            af[i][s] = *reinterpret_cast<const bf16x8*>(sb + mh * C8_HT + a_row0 + i * 2048 + aoff[s]);
      };
      auto readB = [&](int nh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            bfr[nh][j][s] = *reinterpret_cast<const bf16x8*>(sb + (2 + nh) * C8_HT + b_row0 + j * 2048 + aoff[s]);
      };
      auto wait = [&]() {
        if (t > 0) {
          if (draining) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
      };
      auto mma = [&](int mh, int nh) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[mh * 4 + i][nh * 2 + j] =
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nh][j][s], af[i][s], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      };
      readB(0); readA(0); issue(3); wait(); mma(0, 0);
      readB(1); issue(1); wait(); mma(0, 1);
      readA(1); issue(0); wait(); mma(1, 1);
      issue(2); wait(); mma(1, 0);
    }

    // ---- epilogue.  acc[i][j][e]: GEMM row m = m0 + wr*128 + i*16 + r, column n = n0 + wc*64 + j*16 + 4*q4 + e
    int er = r, eq = q4;
    asm volatile("" : "+v"(er), "+v"(eq));
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int mw = tm * 256 + wr * 128, nw = tn * 256 + wc * 64;
    const int ncols = p.N > p.zero_to ? p.N : p.zero_to;
    const bool plain_rows = p.out_stride == 1 && p.out_off == 0 && p.Lgrid == p.Lout;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // everything staged so far has landed before loads / stores join the queue
    if (p.act == ACT_GATED) {
      // gated activation unit (gated_pixelcnn_1d.py:74-83): the weights are interleaved in 32-channel blocks [a | b]
      // (runtime.hip pack_conv), so a wave's 64 GEMM columns are one block: accumulators j = 0, 1 hold a, j = 2, 3 the
      // matching b of the SAME lane; output column = block * 32 + (column inside a)
      const int n_out = p.N >> 1;
      const int ncols_g = n_out > p.zero_to ? n_out : p.zero_to;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = mw + i * 16 + er;
        const bool live = m < p.M;
        int orow = m;
        if (!plain_rows) {
          const int b = m / p.Lgrid;
          orow = b * p.Lout + (m - b * p.Lgrid) * p.out_stride + p.out_off;
        }
        f32x4 gv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool real = nw + j * 16 + 4 * eq < p.N;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // tanh(a) sigmoid(b) on the raw v_exp_f32 / v_rcp_f32: with u = e^(-2|a|), w = e^(-b):
            // sign(a) (1 - u) / ((1 + u)(1 + w)); ~1e-6 relative (1e-7 / |a| near 0), the output is rounded to bf16
            const float a = acc[i][j][e], bb = acc[i][j + 2][e];
            const float u = __builtin_amdgcn_exp2f(-2.885390081777927f * __builtin_fabsf(a));
            const float w = __builtin_amdgcn_exp2f(-1.4426950408889634f * bb);
            float v = (1.f - u) * __builtin_amdgcn_rcpf((1.f + u) * (1.f + w));
            v = a < 0.f ? -v : v;
            gv[j][e] = real ? v : 0.f;
          }
        }
        if (p.C) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int n = (nw >> 1) + j * 16 + 4 * eq;
            if (live && n < ncols_g) *reinterpret_cast<f32x4*>(p.C + (size_t)orow * p.ldc + n) = gv[j];
          }
        }
        if (p.Cb) {
          const unsigned x0 = c8_pack(gv[0][0], gv[0][1]), x1 = c8_pack(gv[0][2], gv[0][3]);
          const unsigned y0 = c8_pack(gv[1][0], gv[1][1]), y1 = c8_pack(gv[1][2], gv[1][3]);
          const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
          const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
          const int n = (nw >> 1) + 16 * (eq & 1) + 8 * (eq >> 1);
          if (live && n < ncols_g) *reinterpret_cast<u32x4*>(p.Cb + (size_t)orow * p.ldcb + n) = v;
        }
        asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3])::"memory");
      }
      continue;
    }
    // bf16 residual rows (indexed by OUTPUT row): the lane's 4 consecutive columns of each accumulator.  The loads are
    // unconditional (rows past M read row M-1, columns past N read column 0: neither is stored) and run one 16-row
    // block ahead of the adds: a load under a branch costs a vmcnt(0), i.e. one memory round trip, per load
    auto out_row = [&](int m) {
      if (plain_rows) return m;
      const int b = m / p.Lgrid;
      return b * p.Lout + (m - b * p.Lgrid) * p.out_stride + p.out_off;
    };
    u32x2 rv[2][4];
    auto load_res = [&](int i, int buf) {
      int m = mw + i * 16 + er;
      m = m < p.M ? m : p.M - 1;
      const unsigned short* rrow = p.ResB + (size_t)out_row(m) * p.ldr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nw + j * 16 + 4 * eq;
        rv[buf][j] = *reinterpret_cast<const u32x2*>(rrow + (n < p.N ? n : 0));
      }
    };
    if (p.ResB) {
      load_res(0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = mw + i * 16 + er;
      const bool live = m < p.M;
      const int orow = out_row(m);
      if (p.ResB) {
        if (i + 1 < 8) load_res(i + 1, (i + 1) & 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x2 x = rv[i & 1][j];
          acc[i][j][0] += __builtin_bit_cast(float, x[0] << 16);
          acc[i][j][1] += __builtin_bit_cast(float, x[0] & 0xffff0000u);
          acc[i][j][2] += __builtin_bit_cast(float, x[1] << 16);
          acc[i][j][3] += __builtin_bit_cast(float, x[1] & 0xffff0000u);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nw + j * 16 + 4 * eq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][e];
          if (p.act == ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
          acc[i][j][e] = n < p.N ? v : 0.f;               // columns [N, zero_to) are written as zeros
        }
      }
      if (p.C) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = nw + j * 16 + 4 * eq;
          if (live && n < ncols) *reinterpret_cast<f32x4*>(p.C + (size_t)orow * p.ldc + n) = acc[i][j];
        }
      }
      if (p.Cb) {
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const unsigned x0 = c8_pack(acc[i][j][0], acc[i][j][1]), x1 = c8_pack(acc[i][j][2], acc[i][j][3]);
          const unsigned y0 = c8_pack(acc[i][j + 1][0], acc[i][j + 1][1]), y1 = c8_pack(acc[i][j + 1][2], acc[i][j + 1][3]);
          const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
          const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
          const int n = nw + 16 * (j + (eq & 1)) + 8 * (eq >> 1);
          if (live && n < ncols) *reinterpret_cast<u32x4*>(p.Cb + (size_t)orow * p.ldcb + n) = v;
        }
      }
      asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3])::"memory");
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();                    // balances the extra barrier waves 4-7 took in the prologue
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the zero block invalid frames read (one per device, allocated at model finalize time, never freed)
static unsigned char* c8_zero[64] = {};

// First use on a device allocates 256 bytes and fills them synchronously: under a lock (two threads may make first
// calls), and at model finalize time for the model path, so that no launch allocates.  The op-level entry point
// pt_conv1d_cl_bf16 calls it too: its FIRST call per device may therefore allocate (documented in the header) and must
// not be made under stream capture.
static std::mutex c8_mu;
int conv8_prepare() {
  int dev = 0;
  PT_HIP(hipGetDevice(&dev));
  PT_CHECK(dev >= 0 && dev < 64, "conv8: device ordinal %d out of range", dev);
  if (__atomic_load_n(&c8_zero[dev], __ATOMIC_ACQUIRE)) return 0;
  std::lock_guard<std::mutex> lk(c8_mu);
  if (!c8_zero[dev]) {
    void* z = nullptr;
    PT_HIP(hipMalloc(&z, 256));
    PT_HIP(hipMemset(z, 0, 256));
    __atomic_store_n(&c8_zero[dev], (unsigned char*)z, __ATOMIC_RELEASE);
  }
  return 0;
}

// Shapes this kernel takes; everything else stays on gemm_bf16_kernel.
bool conv8_takes(const GemmArgsB& a) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !c8_zero[dev]) return false;
  if (a.pe || a.Res) return false;
  if (a.act == ACT_GATED && (a.N % 64 != 0 || a.ResB)) return false;
  if (a.K % 64 != 0 || a.K * a.ntaps < 128 || a.N % 8 != 0 || a.N < 192 || a.N > C8_BIAS_COLS) return false;
  if (a.zero_to % 8 != 0 || (a.Cb && (a.ldcb % 8 != 0 || ((uintptr_t)a.Cb & 15))) || (a.C && (a.ldc % 4 != 0 || ((uintptr_t)a.C & 15))))
    return false;
  if (a.ResB && (a.ldr % 4 != 0 || ((uintptr_t)a.ResB & 7))) return false;
  if ((double)a.Lin * (a.M / a.Lgrid) * a.lda * 2 >= 4.0e9) return false;
  const long tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  return tiles >= 64;                       // fewer: the 4-wave kernel's 128 x 128 tiles fill more CUs
}

int launch_conv8(const GemmArgsB& a, hipStream_t s) {
  int dev = 0;
  PT_HIP(hipGetDevice(&dev));
  PT_CHECK(conv8_takes(a), "conv8: shape not supported (internal dispatch error)");
  PT_ENSURE_LDS(conv8_kernel, C8_SMEM);
  const int c8_num_cus = device_cus();
  const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256, ntiles = tiles_m * tiles_n;
  const int grid = ntiles < c8_num_cus ? ntiles : c8_num_cus;
  void* tok = nullptr;
  if (prof_is_on()) prof_record_begin(s, a.flops > 0 ? a.flops : 2.0 * a.M * a.N * a.K * a.ntaps, 3, &tok);
  PT_LAUNCH(conv8_kernel, dim3(grid), dim3(512), C8_SMEM, s, a, (const unsigned char*)c8_zero[dev], tiles_n, ntiles);
  if (tok) prof_record_end(s, tok);
  return 0;
}

}  // namespace pt
