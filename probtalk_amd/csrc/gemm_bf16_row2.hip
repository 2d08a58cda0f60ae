// Row-complete bf16 GEMM for the d_model = 512 Linears, TWO workgroups per CU (round 3).
//
// gemm_bf16_row.hip runs one persistent 8-wave workgroup per CU on 160-row tiles: its LayerNorm epilogues (VALU +
// HBM round trips: 10 us of a 25 us LNMOD tile, 24 us of a 40 us ADDLN tile) stand beside an idle matrix pipe, because
// all eight waves of the CU are in the epilogue at the same time.  This variant halves the footprint of a workgroup
// (80-row tiles: 80 accumulator registers per lane, 79 KB of LDS) so that two workgroups share a CU, four waves per
// SIMD: while one normalises and stores its rows the other's MFMAs keep the matrix pipe busy (the vector and the
// matrix pipes of a SIMD run side by side), and a workgroup that finishes is replaced by the next tile of the grid, so
// a launch has no lock-step rounds and a second stream's kernels can move into the CUs a launch leaves.
//
//   tile BM x 512, BM = 16*MT (MT = 5); 8 waves as 1 (M) x 8 (N), wave wn owns columns [64 wn, 64 wn + 64) of all rows:
//   MT x 4 accumulators of v_mfma_f32_16x16x32_bf16 with swapped operands (a lane owns one row, 4 columns per accumulator);
//   BK = 32.  LDS: A ring of three K-tiles (BM rows x 64 B each) + W ring of two K-tiles ([B-lo 256 rows | B-hi 256 rows]
//   x 64 B = 32 KB each), filled by LDS-DMA; the LayerNorm tables and the row-statistics exchange of the epilogue reuse
//   the W stage the last K-tile leaves free.
//   K-tile t:  s_waitcnt (W(t) and A(t) landed; A(t+1) may be in flight)  s_barrier
//              issue W(t+1), A(t+2)   [A runs two K-tiles ahead: its rows come from HBM, W from L2]
//              fragment reads, 4 x MT MFMAs.
//   One grid workgroup per tile (no persistence: the second workgroup of the CU covers prologue and epilogue).
// Same LDS image layout, source-side swizzle, accumulator layout and epilogue arithmetic as gemm_bf16_row.hip.
#include <cstdlib>

#include "pt_common.h"

namespace pt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

void prof_record_begin(hipStream_t s, double flops, int cfg, void** tok);
void prof_record_end(hipStream_t s, void* tok);
bool prof_is_on();

constexpr int R2_WSTAGE = 32768;                  // [B-lo | B-hi]
constexpr int R2_TAB = 0;                         // inside the free W stage: LayerNorm tables, 4 classes x 4 KB at most
constexpr int R2_RED = 16384;                     // inside the free W stage: [2][8 waves][160] floats
constexpr int R2_TAB_CLASSES = 4;
constexpr int r2_smem(int mt) { return 3 * mt * 1024 + 2 * R2_WSTAGE; }

__device__ __forceinline__ unsigned r2_pack_bf16(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

constexpr int EPI2_ADDLN_B = 3;   // EPI_ADDLN with the bf16 residual stream (X == nullptr)

template <int MT, int EPIX>
__global__ __launch_bounds__(512, 4) void gemm_row2_kernel(const GemmBigArgs p) {
  constexpr int BM = 16 * MT;
  constexpr int ASLOT = MT * 1024;
  constexpr bool RESB = EPIX == EPI2_ADDLN_B;
  constexpr int EPI = RESB ? EPI_ADDLN : EPIX;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const wst = smem + 3 * ASLOT;
  const int tile = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q4 = lane >> 4;
  const int nt = p.K >> 5;
  const bool hasA = wave < MT;                        // this wave stages 16 rows of the A image

  // ---- DMA sources.  One wave-instruction = 16 image rows x 64 B; lane -> row (lane >> 2), LDS chunk (lane & 3)
  auto gsw = [](int k) { return (0x78 >> (2 * k)) & 3; };   // g[0]=0, g[1]=2, g[2]=3, g[3]=1
  unsigned soffA, soffL[2], soffH[2];
  {
    const int lr = wave * 16 + (lane >> 2);
    const int g = (lane & 3) ^ gsw((lr >> 2) & 3);
    int m = tile * BM + (hasA ? lr : 0);
    m = m < p.M ? m : p.M - 1;
    soffA = ((unsigned)m * (unsigned)p.lda + g * 8) * 2u;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = (wave + 8 * i) * 16 + (lane >> 2);          // row of the B-lo / B-hi image: wave (lr >> 5), row (lr & 31)
    const int g = (lane & 3) ^ gsw((lr >> 2) & 3);
    const int n = (lr >> 5) * 64 + (lr & 31);
    soffL[i] = ((unsigned)n * (unsigned)p.K + g * 8) * 2u;
    soffH[i] = ((unsigned)(n + 32) * (unsigned)p.K + g * 8) * 2u;
  }
  auto issueA = [&](int kt) {
    if (hasA) {
      const unsigned char* ab = reinterpret_cast<const unsigned char*>(p.A) + (size_t)kt * 64;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ab + soffA), (lds_ptr_t)(smem + (kt % 3) * ASLOT + wave * 1024), 16, 0, 0);
    }
  };
  auto issueW = [&](int kt) {
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.W) + (size_t)kt * 64;
    unsigned char* st = wst + (kt & 1) * R2_WSTAGE;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + soffL[0]), (lds_ptr_t)(st + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + soffL[1]), (lds_ptr_t)(st + (wave + 8) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + soffH[0]), (lds_ptr_t)(st + 16384 + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + soffH[1]), (lds_ptr_t)(st + 16384 + (wave + 8) * 1024), 16, 0, 0);
  };

  const int foff = r * 64 + ((q4 ^ gsw((r >> 2) & 3)) << 4);     // fragment read offset inside a 16-row block
  const int bro = wave * 32 * 64;                                // this wave's rows of the B-lo / B-hi image

  f32x4 acc[MT][4];
  // accumulators start at the bias
  {
    const int nb = wave * 64 + 4 * q4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nb + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = bv;
    }
  }
  issueA(0);
  issueW(0);
  if (nt > 1) issueA(1);

  // the W stage that the last K-tile does not read: tables and row statistics of the epilogue live there
  unsigned char* const fre = wst + (nt & 1) * R2_WSTAGE;
  float* const tab = reinterpret_cast<float*>(fre + R2_TAB);
  float* const red = reinterpret_cast<float*>(fre + R2_RED);

#define PT_PIN_BLOCK2(i) asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3])::"memory")

  for (int t = 0; t < nt; ++t) {
    // W(t) and A(t) have landed; the A instruction of K-tile t+1 (younger than W(t)) may stay in flight
    if (hasA && t + 1 < nt) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nt) {
      issueW(t + 1);
      if (t + 2 < nt) issueA(t + 2);
    } else {
      // last K-tile: the other W stage is free (read during K-tile nt-2, every wave is past this barrier): the
      // LayerNorm tables go there by DMA, linear copies of 1 KB per instruction
      if constexpr (EPI == EPI_LNMOD) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int piece = wave * 2 + i;                        // 4 pieces of 256 floats per class
          if (piece < p.n_classes * 4)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(p.style + piece * 256 + lane * 4), (lds_ptr_t)(fre + R2_TAB + piece * 1024), 16, 0, 0);
        }
      } else if constexpr (EPI == EPI_ADDLN) {
        if (wave < 2) __builtin_amdgcn_global_load_lds((glb_ptr_t)(p.gamma + wave * 256 + lane * 4), (lds_ptr_t)(fre + R2_TAB + wave * 1024), 16, 0, 0);
        else if (wave < 4) __builtin_amdgcn_global_load_lds((glb_ptr_t)(p.beta + (wave - 2) * 256 + lane * 4), (lds_ptr_t)(fre + R2_TAB + wave * 1024), 16, 0, 0);
      }
    }
    const unsigned char* sa = smem + (t % 3) * ASLOT;
    const unsigned char* sw = wst + (t & 1) * R2_WSTAGE + bro;
    bf16x8 af[MT], bl[2], bh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bl[j] = *reinterpret_cast<const bf16x8*>(sw + j * 1024 + foff);
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 1024 + foff);
#pragma unroll
    for (int j = 0; j < 2; ++j) bh[j] = *reinterpret_cast<const bf16x8*>(sw + 16384 + j * 1024 + foff);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], af[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], af[i], acc[i][2 + j], 0, 0, 0);
  }

  // ---- epilogue.  acc[i][j][e]: row m = tile*BM + i*16 + er, column n = wave*64 + j*16 + 4*eq + e
  int er = r, eq = q4;
  asm volatile("" : "+v"(er), "+v"(eq));
  const int m_base = tile * BM, n_base = wave * 64 + 4 * eq;
  int cls[MT];
  if constexpr (EPI == EPI_LNMOD) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m_base + i * 16 + er;
      const int mc = m < p.M ? m : p.M - 1;
      const long long lb = p.label[p.n_label == 1 ? 0 : mc / p.rows_per_clip];
      cls[i] = lb < 0 ? 0 : (lb >= p.n_classes ? p.n_classes - 1 : (int)lb);
    }
  }
  (void)cls;
  if constexpr (EPI == EPI_ADDLN) {
    if constexpr (!RESB) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        int m = m_base + i * 16 + er;
        m = m < p.M ? m : p.M - 1;                       // unconditional loads; row M-1 is never stored by a dead lane
        f32x4 xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * 512 + n_base + j * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += xv[j];
        PT_PIN_BLOCK2(i);
      }
    } else {
      // bf16 residual stream: the residual is the bf16 row this epilogue overwrites, loaded in the store layout (a lane
      // reads the 8 consecutive bf16 it will write) and un-paired with the same v_permlane16_swap; one 16-row block per
      // group, two groups in flight (128 registers per lane: two workgroups per CU)
      constexpr int XB = 1, NG = (MT + XB - 1) / XB;
      u32x4 xr[2][XB][2];
      auto load_grp = [&](int g, int buf) {
#pragma unroll
        for (int k = 0; k < XB; ++k) {
          if (g * XB + k < MT) {
            int m = m_base + (g * XB + k) * 16 + er;
            m = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              // 32-bit byte offset from the uniform base (operands are below 4 GB): one address register per load
              const unsigned n = wave * 64 + 16 * (2 * jj + (eq & 1)) + 8 * (eq >> 1);
              const unsigned off = ((unsigned)m * (unsigned)p.ldcb + n) * 2u;
              xr[buf][k][jj] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.Cb) + off);
            }
          }
        }
      };
      auto add_grp = [&](int g, int buf) {
#pragma unroll
        for (int k = 0; k < XB; ++k) {
          if (g * XB + k < MT) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const u32x4 x = xr[buf][k][jj];
              const auto a0 = __builtin_amdgcn_permlane16_swap(x[0], x[2], false, false);   // -> x0, y0
              const auto a1 = __builtin_amdgcn_permlane16_swap(x[1], x[3], false, false);   // -> x1, y1
              f32x4& lo = acc[g * XB + k][2 * jj];
              f32x4& hi = acc[g * XB + k][2 * jj + 1];
              lo[0] += __builtin_bit_cast(float, a0[0] << 16);
              lo[1] += __builtin_bit_cast(float, a0[0] & 0xffff0000u);
              lo[2] += __builtin_bit_cast(float, a1[0] << 16);
              lo[3] += __builtin_bit_cast(float, a1[0] & 0xffff0000u);
              hi[0] += __builtin_bit_cast(float, a0[1] << 16);
              hi[1] += __builtin_bit_cast(float, a0[1] & 0xffff0000u);
              hi[2] += __builtin_bit_cast(float, a1[1] << 16);
              hi[3] += __builtin_bit_cast(float, a1[1] & 0xffff0000u);
            }
            PT_PIN_BLOCK2(g * XB + k);
          }
        }
      };
      load_grp(0, 0);
      if constexpr (NG > 1) load_grp(1, 1);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        add_grp(g, g & 1);
        if (g + 2 < NG) load_grp(g + 2, g & 1);
      }
    }
  }
  float* outf = RESB ? nullptr : EPI == EPI_ADDLN ? p.X : (EPI == EPI_STD ? p.C : nullptr);
  const int ldf = EPI == EPI_ADDLN ? 512 : p.ldc;
  auto store_block = [&](int i) {
    const int m = m_base + i * 16 + er;
    const bool live = m < p.M;
    if (outf && live) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(outf + (size_t)m * ldf + n_base + j * 16) = acc[i][j];
    }
    if (p.Cb) {
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const unsigned x0 = r2_pack_bf16(acc[i][j][0], acc[i][j][1]), x1 = r2_pack_bf16(acc[i][j][2], acc[i][j][3]);
        const unsigned y0 = r2_pack_bf16(acc[i][j + 1][0], acc[i][j + 1][1]), y1 = r2_pack_bf16(acc[i][j + 1][2], acc[i][j + 1][3]);
        const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
        const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
        const unsigned n = wave * 64 + 16 * (j + (eq & 1)) + 8 * (eq >> 1);
        const unsigned off = ((unsigned)m * (unsigned)p.ldcb + n) * 2u;
        if (live) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(p.Cb) + off) = v;
      }
    }
  };
  if constexpr (EPI != EPI_STD) {
    // LayerNorm over the 512 columns of a row: lane partials -> 4 lanes of a row -> 8 waves through LDS (two passes:
    // mean, then the sum of squared deviations)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc[i][j][e];
      s = row_lanes_sum(s);
      if (eq == 0) red[wave * 160 + i * 16 + er] = s;
      PT_PIN_BLOCK2(i);
    }
    // the partial sums must be IN the LDS before another wave reads them: a bare s_barrier does not wait for this
    // wave's outstanding ds_write (the last block's sums were read stale, rarely, with four waves per SIMD)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float mean_[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float* pr = red + 2 * eq * 160 + i * 16 + er;
      mean_[i] = row_lanes_sum(pr[0] + pr[160]) * (1.f / 512.f);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float mean = mean_[i];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = acc[i][j][e] - mean;
          s += d * d;
        }
      s = row_lanes_sum(s);
      if (eq == 0) red[8 * 160 + wave * 160 + i * 16 + er] = s;
      PT_PIN_BLOCK2(i);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the table DMA of the last K-tile has landed, the sums are in the LDS
    __builtin_amdgcn_s_barrier();
    // lane coordinates laundered again: the store offsets are recomputed here instead of living in registers (and in
    // scratch) from the residual loads on
    asm volatile("" : "+v"(er), "+v"(eq));
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      // (the eight per-wave partials of the row: two per lane, the lane sum does the rest; read per block here, the
      // other waves of the SIMD cover the LDS round trip and five registers stay free)
      const float* pr = red + 8 * 160 + 2 * eq * 160 + i * 16 + er;
      const float mean = mean_[i], rstd = __builtin_amdgcn_rsqf(row_lanes_sum(pr[0] + pr[160]) * (1.f / 512.f) + 1e-5f);
      const float nmr = -mean * rstd;
      const float* gp = tab + n_base;
      if constexpr (EPI == EPI_LNMOD) gp += cls[i] * 1024;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gp + j * 16);
        const f32x4 be = *reinterpret_cast<const f32x4*>(gp + 512 + j * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float y = __builtin_fmaf(__builtin_fmaf(acc[i][j][e], rstd, nmr), ga[e], be[e]);
          if (EPI == EPI_LNMOD) y = y * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
          acc[i][j][e] = y;
        }
      }
      store_block(i);
      PT_PIN_BLOCK2(i);
    }
  } else {
    if (p.relu) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] = acc[i][j][e] > 0.f ? acc[i][j][e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      store_block(i);
      PT_PIN_BLOCK2(i);
    }
  }
#undef PT_PIN_BLOCK2
}

template <int MT, int EPI>
static int launch_row2_t(const GemmBigArgs& a, hipStream_t s) {
  PT_ENSURE_LDS((gemm_row2_kernel<MT, EPI>), r2_smem(MT));
  const int ntiles = (a.M + 16 * MT - 1) / (16 * MT);
  void* tok = nullptr;
  if (prof_is_on()) prof_record_begin(s, a.flops > 0 ? a.flops : 2.0 * a.M * a.N * a.K, 7, &tok);
  PT_LAUNCH((gemm_row2_kernel<MT, EPI>), dim3(ntiles), dim3(512), r2_smem(MT), s, a);
  if (tok) prof_record_end(s, tok);
  return 0;
}

// Arguments as launch_gemm_row's (N = Npad = 512, K a multiple of 32, no pe / Res / zero_to); LNMOD with at most 4 classes.
int launch_gemm_row2(const GemmBigArgs& a, hipStream_t s) {
  PT_CHECK(a.K % 32 == 0 && a.K >= 64, "gemm_row2: K=%d must be a multiple of 32, at least 64", a.K);
  PT_CHECK(a.epi != EPI_LNMOD || a.n_classes <= R2_TAB_CLASSES, "gemm_row2: at most %d style classes", R2_TAB_CLASSES);
  if (a.epi == EPI_STD) return launch_row2_t<5, EPI_STD>(a, s);
  if (a.epi == EPI_LNMOD) return launch_row2_t<5, EPI_LNMOD>(a, s);
  if (a.X) return launch_row2_t<5, EPI_ADDLN>(a, s);
  return launch_row2_t<5, EPI2_ADDLN_B>(a, s);
}

}  // namespace pt
