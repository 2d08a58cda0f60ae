// Row-complete bf16 GEMM for every d_model = 512 Linear of the transformers: C[M,512] = A[M,K] . W[512,K]^T with the
// LayerNorm that follows the Linear fused into the epilogue (a workgroup owns whole rows).  Same phase-pipelined
// structure as gemm_bf16_8ph.hip, re-shaped for N = 512:
//
//   tile BM x 512, BM = 16*MT (MT = 8 or 10: 128 or 160 rows, chosen per launch so that the tile count fills whole rounds
//   of the 256 CUs: 75,776 rows are 592 tiles of 128 = 3 rounds, but 474 tiles of 160 = 2 rounds);
//   8 waves as 1 (M) x 8 (N): wave wn owns columns [64 wn, 64 wn + 64) of ALL BM rows = MT x 4 accumulators of
//   v_mfma_f32_16x16x32_bf16 (operands swapped: a lane owns one row and 4 consecutive columns per accumulator);
//   BK = 32 (one MFMA k-step), three LDS stages of [A: BM rows | B-lo: 256 rows | B-hi: 256 rows] x 64 B filled by LDS-DMA;
//   B-lo / B-hi = the W rows of every wave's column half nh = 0 / 1.
// A K-tile is two phases, each [ds_reads | DMA issue | counted vmcnt] s_barrier [MT x 2 MFMAs] s_barrier:
//   P1 reads A + B-lo, issues B-lo of K-tile t+2, MFMAs column half 0;  P2 reads B-hi, issues A then B-hi of t+2, half 1.
// Waves 4-7 run one barrier behind waves 0-3 (SIMD partners alternate read/DMA and MFMA segments).  A stage is
// re-filled two phases after its last read, retired by the counted wait three phases after its issue and read four
// phases after its issue.  The K-tile stream runs across tiles (persistent workgroups).
// LDS rows are 64 B: 16-byte chunk c of row r sits at chunk c ^ g[(r >> 2) & 3], g = {0,2,3,1} (source-side swizzle,
// conflict-free for the four 16-lane groups of a ds_read_b128).
//
// Epilogues (pt_common.h): EPI_STD (bias / ReLU), EPI_LNMOD, EPI_ADDLN.  No LDS slabs: fp32 rows leave as 16 rows x 64 B
// per store instruction straight from the accumulators; bf16 rows are paired across adjacent accumulators with
// v_permlane16_swap so that a lane stores 16 B and an instruction again covers 16 rows x 64 B.
#include <cstdlib>

#include "pt_common.h"

namespace pt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

void prof_record_begin(hipStream_t s, double flops, int cfg, void** tok);
void prof_record_end(hipStream_t s, void* tok);
bool prof_is_on();

constexpr int GR_AMAX = 10 * 1024;                 // A image of the largest tile (160 rows x 64 B)
constexpr int GR_STAGE = GR_AMAX + 32768;          // [A | B-lo | B-hi]
constexpr int GR_RED = 3 * GR_STAGE;               // LayerNorm partial sums: [2][8 waves][160 rows] floats
constexpr int GR_TAB = GR_RED + 2 * 8 * 160 * 4;   // LayerNorm affine tables: [classes][gamma' 512 | beta' 512] floats, 4 classes at most
constexpr int GR_TAB_CLASSES = 4;
constexpr int GR_BIAS = GR_TAB + GR_TAB_CLASSES * 4096;   // the Linear's bias row (512 floats)
constexpr int GR_SMEM = GR_BIAS + 2048;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

constexpr int EPI_ADDLN_B = 3;   // EPI_ADDLN with the bf16 residual stream (X == nullptr): its own instantiation

template <int MT, int EPIX>
__global__ __launch_bounds__(512) void gemm_row_kernel(const GemmBigArgs p, const int ntiles) {
  constexpr int BM = 16 * MT;
  constexpr bool RESB = EPIX == EPI_ADDLN_B;
  constexpr int EPI = RESB ? EPI_ADDLN : EPIX;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int G = gridDim.x;
  const int tile0 = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                         // half: 0 = waves 0-3, 1 = waves 4-7 (one barrier behind)
  const int r = lane & 15, q4 = lane >> 4;
  const int nt = p.K >> 5;
  const bool extra = MT > 8 && wave + 8 < MT;        // this wave issues a second A instruction per K-tile

  // ---- DMA streams.  One wave-instruction = 16 image rows x 64 B; lane -> row (lane >> 2), LDS chunk (lane & 3).
  auto gsw = [](int k) { return (0x78 >> (2 * k)) & 3; };   // 0x78 = 01 11 10 00b -> g[0]=0, g[1]=2, g[2]=3, g[3]=1
  // W either row-major [512][K] (64 bytes of every row per K-tile: half cache lines) or K-tile-major [K / 32][512][32]
  // (p.Wt: a K-tile is 32 KB of consecutive memory, 2-6 % faster per launch)
  const unsigned char* const Wsrc = reinterpret_cast<const unsigned char*>(p.Wt ? p.Wt : p.W);
  const unsigned wrow = p.Wt ? 64u : (unsigned)p.K * 2u;
  const size_t wkt = p.Wt ? 32768 : 64;
  unsigned soffA[2], soffL[2], soffH[2];
  auto set_srcA = [&](int tl) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (wave + 8 * i) * 16 + (lane >> 2);
      const int g = (lane & 3) ^ gsw((lr >> 2) & 3);
      int m = tl * BM + lr;
      m = m < p.M ? m : p.M - 1;
      soffA[i] = ((unsigned)m * (unsigned)p.lda + g * 8) * 2u;
    }
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = (wave + 8 * i) * 16 + (lane >> 2);          // row of the B-lo / B-hi image: wave (lr >> 5), row (lr & 31)
    const int g = (lane & 3) ^ gsw((lr >> 2) & 3);
    const int n = (lr >> 5) * 64 + (lr & 31);
    soffL[i] = (unsigned)n * wrow + g * 16u;
    soffH[i] = (unsigned)(n + 32) * wrow + g * 16u;
  }
  // cursors: 0 = B-lo (issued in P1, beside the 12 fragment reads), 1 = A + B-hi (P2, which reads 2 fragments only)
  int c_tile[2], c_kt[2], c_buf[2];
  bool draining = false;
  auto advance = [&](int c) {
    c_buf[c] = c_buf[c] == 2 ? 0 : c_buf[c] + 1;
    if (++c_kt[c] == nt) {
      c_kt[c] = 0;
      c_tile[c] += G;
      if (c == 1 && c_tile[c] < ntiles) set_srcA(c_tile[c]);
    }
  };
  auto issue_p1 = [&]() {
    if (c_tile[0] >= ntiles) {
      draining = true;
      return;
    }
    const unsigned char* wb = Wsrc + (size_t)c_kt[0] * wkt;
    unsigned char* st = smem + c_buf[0] * GR_STAGE;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + soffL[0]), (lds_ptr_t)(st + GR_AMAX + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + soffL[1]), (lds_ptr_t)(st + GR_AMAX + (wave + 8) * 1024), 16, 0, 0);
    advance(0);
  };
  auto issue_p2 = [&]() {
    if (c_tile[1] >= ntiles) {
      draining = true;
      return;
    }
    const unsigned char* wb = Wsrc + (size_t)c_kt[1] * wkt;
    const unsigned char* ab = reinterpret_cast<const unsigned char*>(p.A) + (size_t)c_kt[1] * 64;
    unsigned char* sa = smem + c_buf[1] * GR_STAGE;
    unsigned char* st = sa + GR_AMAX + 16384;
    // A first: the P2 wait one K-tile later retires it (it is read in the following P1), B-hi may still be in flight then
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(ab + soffA[0]), (lds_ptr_t)(sa + wave * 1024), 16, 0, 0);
    if (extra) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ab + soffA[1]), (lds_ptr_t)(sa + (wave + 8) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + soffH[0]), (lds_ptr_t)(st + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + soffH[1]), (lds_ptr_t)(st + (wave + 8) * 1024), 16, 0, 0);
    advance(1);
  };
  // LayerNorm affine tables into LDS (read per 16-row block in the epilogue: from global memory each block paid a
  // label -> table -> value chain of dependent loads)
  float* tab = reinterpret_cast<float*>(smem + GR_TAB);
  if constexpr (EPI == EPI_LNMOD) {
    for (int k = tid; k < p.n_classes * 256; k += 512)
      *reinterpret_cast<f32x4*>(tab + 4 * k) = *reinterpret_cast<const f32x4*>(p.style + 4 * k);
  } else if constexpr (EPI == EPI_ADDLN) {
    if (tid < 128) *reinterpret_cast<f32x4*>(tab + 4 * tid) = *reinterpret_cast<const f32x4*>(p.gamma + 4 * tid);
    else if (tid < 256) *reinterpret_cast<f32x4*>(tab + 4 * tid) = *reinterpret_cast<const f32x4*>(p.beta + 4 * (tid - 128));
  }
  // ... and the bias row: read from memory at the head of every tile it made the compiler wait for vmcnt(0) before the
  // tile's first MFMA, i.e. for every store of the tile before (round 3, profiles/r3/tile_head_bias_loads.txt)
  float* bias_s = reinterpret_cast<float*>(smem + GR_BIAS);
  if (tid < 128) *reinterpret_cast<f32x4*>(bias_s + 4 * tid) = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + 4 * tid) : f32x4{0.f, 0.f, 0.f, 0.f};
  c_tile[0] = c_tile[1] = tile0;
  c_kt[0] = c_kt[1] = 0;
  c_buf[0] = c_buf[1] = 0;
  set_srcA(tile0);
  issue_p1(); issue_p2(); issue_p1(); issue_p2();                // K-tiles 0 and 1
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();                    // waves 4-7 run one barrier behind

  const int foff = r * 64 + ((q4 ^ gsw((r >> 2) & 3)) << 4);     // fragment read offset inside a 16-row block
  const int bro = wave * 32 * 64;                                // this wave's rows of the B-lo / B-hi image

  f32x4 acc[MT][4];
  bf16x8 af[MT], bfr[2];
  int cbuf = 0;
  float* red = reinterpret_cast<float*>(smem + GR_RED);


  // keeps one 16-row block's values in place: without it the compiler hoists every epilogue load above the arithmetic
  // of all blocks and spills hundreds of registers
#define PT_PIN_BLOCK(i)                                                   \
  asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3])::"memory")
  for (int tile = tile0; tile < ntiles; tile += G) {
    // accumulators start at the bias.  (Starting them at bias + residual row for EPI_ADDLN was measured: the 320 KB of x per
    // tile then stand in front of the first MFMAs, 142 us against 130 us per launch at 75,776 rows.)
    {
      int ir = r, iq = q4;
      asm volatile("" : "+v"(ir), "+v"(iq));
      const int nb = wave * 64 + 4 * iq;
      f32x4 bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bias_s + nb + j * 16);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = tile * BM + i * 16 + ir;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = bv[j];
        }
        (void)m;
      }
    }

    for (int t = 0; t < nt; ++t) {
      const unsigned char* sb = smem + cbuf * GR_STAGE;
      cbuf = cbuf == 2 ? 0 : cbuf + 1;
      auto mma = [&](int nh) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i)
            acc[i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][nh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      };
      // P1
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + GR_AMAX + bro + j * 1024 + foff);
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + i * 1024 + foff);
      issue_p1();
      if (t >= 2) {                                              // retires A and B-hi issued 3 phases ago (read from P2 / next P1 on)
        if (draining) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (extra) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // B-lo 2 + (A 2 + B-hi 2) + B-lo 2
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                  // B-lo 2 + (A 1 + B-hi 2) + B-lo 2
      }
      mma(0);
      // P2
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + GR_AMAX + 16384 + bro + j * 1024 + foff);
      issue_p2();
      if (t >= 1) {                                              // retires B-lo issued 3 phases ago AND the A issued 2 phases ago
        if (draining) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (extra) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // B-hi 2 + B-lo 2 + (A 2 + B-hi 2)
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                  // B-hi 2 + B-lo 2 + (A 1 + B-hi 2)
      }
      mma(1);
    }

    // ---- epilogue.  acc[i][j][e]: row m = tile*BM + i*16 + er, column n = wave*64 + j*16 + 4*eq + e
    // (lane coordinates are laundered through an empty asm so that the epilogue's address arithmetic is not hoisted out
    // of the tile loop, where it would cost the main loop its registers)
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
        // fp32 residual rows, two 16-row blocks per round trip (the fragment registers are free here)
#pragma unroll
        for (int i = 0; i < MT; i += 2) {
          f32x4 xv[2][4];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            int m = m_base + (i + k) * 16 + er;
            m = m < p.M ? m : p.M - 1;                       // unconditional loads (see the bf16 branch); row M-1 is never stored
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[k][j] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * 512 + n_base + j * 16);
          }
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i + k][j] += xv[k][j];
          PT_PIN_BLOCK(i);
          PT_PIN_BLOCK(i + 1);
        }
      } else {
        // bf16 residual stream (X == nullptr): the residual is the bf16 row this epilogue overwrites.  Loaded in the
        // store layout (a lane reads the 8 consecutive bf16 it will write: in place is safe) and un-paired with the
        // same v_permlane16_swap.  Groups of two 16-row blocks, two groups in flight: group g+2 is requested as soon as
        // group g has been added, so the loads stream instead of paying one round trip per group.
        constexpr int XB = 2, NG = (MT + XB - 1) / XB;
        u32x4 xr[2][XB][2];
        auto load_grp = [&](int g, int buf) {
#pragma unroll
          for (int k = 0; k < XB; ++k) {
            if (g * XB + k < MT) {
              // rows past M read row M-1 (never stored): an unconditional load keeps the group free of branches, and a
              // branch per load costs a vmcnt(0), i.e. one memory round trip, per load
              int m = m_base + (g * XB + k) * 16 + er;
              m = m < p.M ? m : p.M - 1;
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) {
                const int n = wave * 64 + 16 * (2 * jj + (eq & 1)) + 8 * (eq >> 1);
                xr[buf][k][jj] = *reinterpret_cast<const u32x4*>(p.Cb + (size_t)m * p.ldcb + n);
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
              PT_PIN_BLOCK(g * XB + k);
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
    // one 16-row block: fp32 rows straight from the accumulators (16 rows x 64 B per instruction); bf16: accumulators j
    // and j+1 exchange 16-lane rows (v_permlane16_swap), afterwards a lane holds 8 consecutive bf16 columns
    // (column 16*(j + (eq & 1)) + 8*(eq >> 1) of the wave's 64)
    float* outf = RESB ? nullptr : EPI == EPI_ADDLN ? p.X : (EPI == EPI_STD ? p.C : nullptr);   // ADDLN with a bf16 residual stream: none
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
          const unsigned x0 = pack_bf16(acc[i][j][0], acc[i][j][1]), x1 = pack_bf16(acc[i][j][2], acc[i][j][3]);
          const unsigned y0 = pack_bf16(acc[i][j + 1][0], acc[i][j + 1][1]), y1 = pack_bf16(acc[i][j + 1][2], acc[i][j + 1][3]);
          const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
          const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
          const int n = wave * 64 + 16 * (j + (eq & 1)) + 8 * (eq >> 1);
          if (live) *reinterpret_cast<u32x4*>(p.Cb + (size_t)m * p.ldcb + n) = v;
        }
      }
    };
    if constexpr (EPI != EPI_STD) {
      // LayerNorm over the 512 columns of a row: lane partials -> 4 lanes of a row (eq) -> 8 waves through LDS.
      // The partial sums cross the two staggered halves, so waves 0-3 wait for waves 4-7 here and the stagger is
      // re-established after the epilogue.
      if (grp == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) s += acc[i][j][e];
        s = row_lanes_sum(s);
        if (eq == 0) red[wave * 160 + i * 16 + er] = s;
        PT_PIN_BLOCK(i);
      }
      // the partial sums must be IN the LDS before another wave reads them: a bare s_barrier does not wait for this
      // wave's outstanding ds_write
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // the 8 per-wave partials of a row: each of the row's 4 lanes reads two of them (one ds_read2), the lane sum does
      // the rest.  All MT blocks' reads go out in one batch (independent, one wait) instead of one dependent LDS round
      // trip per block; the means stay in registers for the last pass
      float mean_[MT], rstd_[MT];
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
        PT_PIN_BLOCK(i);
      }
      // every K-tile staged so far has landed before the stores join the in-order vmcnt queue (and the sums are in the LDS)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float* pr = red + 8 * 160 + 2 * eq * 160 + i * 16 + er;
        rstd_[i] = __builtin_amdgcn_rsqf(row_lanes_sum(pr[0] + pr[160]) * (1.f / 512.f) + 1e-5f);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float mean = mean_[i], rstd = rstd_[i];
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
            // SiLU = y / (1 + 2^(-y log2 e)) on the raw v_exp_f32 / v_rcp_f32 (2^z overflows to inf -> y * 0 for y << 0)
            if (EPI == EPI_LNMOD) y = y * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
            acc[i][j][e] = y;
          }
        }
        store_block(i);
        PT_PIN_BLOCK(i);
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
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        store_block(i);
        PT_PIN_BLOCK(i);
      }
    }
    if constexpr (EPI != EPI_STD) {
      if (grp == 1) __builtin_amdgcn_s_barrier();                // waves 4-7 fall one barrier behind again
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();                    // balances the extra barrier waves 4-7 took in the prologue
}


template <int MT, int EPI>
static int launch_row(const GemmBigArgs& a, hipStream_t s) {
  PT_ENSURE_LDS((gemm_row_kernel<MT, EPI>), GR_SMEM);
  const int gr_num_cus = device_cus();
  const int ntiles = (a.M + 16 * MT - 1) / (16 * MT);
  const int grid = ntiles < gr_num_cus ? ntiles : gr_num_cus;
  void* tok = nullptr;
  if (prof_is_on()) prof_record_begin(s, a.flops > 0 ? a.flops : 2.0 * a.M * a.N * a.K, 7, &tok);
  PT_LAUNCH((gemm_row_kernel<MT, EPI>), dim3(grid), dim3(512), GR_SMEM, s, a, ntiles);
  if (tok) prof_record_end(s, tok);
  return 0;
}

int launch_gemm_row2(const GemmBigArgs& a, hipStream_t s);

// Tile height: the one whose tile count wastes less of the last round of the persistent grid.
int launch_gemm_row(const GemmBigArgs& a_in, hipStream_t s) {
  GemmBigArgs a = a_in;
  if (option(OPT_W_KTILE) != 1) a.Wt = nullptr;
  const int gr_num_cus = device_cus();
  const int force_mt = option(OPT_ROW_TILE);
  // enough rows for more 80-row tiles than CUs: two workgroups per CU, epilogues beside the other workgroup's MFMAs
  if (!force_mt && option(OPT_ROW2) == 1 && a.M >= option(OPT_ROW2_MIN_ROWS) && (a.epi != EPI_LNMOD || a.n_classes <= GR_TAB_CLASSES))
    return launch_gemm_row2(a, s);
  auto cost = [&](int bm) {
    const long tiles = (a.M + bm - 1) / bm;
    return ((tiles + gr_num_cus - 1) / gr_num_cus) * (long)bm;
  };
  int mt = force_mt ? force_mt : (cost(160) <= cost(128) ? 10 : 8);       // ties: the taller tile re-reads W less often
  if (!force_mt) {
    // fewer tiles than CUs: the launch is one round whatever the tile height, and its length is the K loop of one tile
    // (latency per K-tile, not MFMA work), so shorter tiles on more CUs finish sooner
    if ((a.M + 63) / 64 <= gr_num_cus) mt = 4;
    if ((a.M + 31) / 32 <= gr_num_cus) mt = 2;
  }
  PT_CHECK(mt == 2 || mt == 4 || mt == 8 || mt == 10, "gemm_row: tile height 16*%d not built (option row_tile = 2, 4, 8 or 10)", mt);
#define PT_ROW_DISPATCH(E)                                            \
  switch (mt) {                                                       \
    case 2: return launch_row<2, E>(a, s);                            \
    case 4: return launch_row<4, E>(a, s);                            \
    case 8: return launch_row<8, E>(a, s);                            \
    default: return launch_row<10, E>(a, s);                          \
  }
  if (a.epi == EPI_STD) PT_ROW_DISPATCH(EPI_STD)
  if (a.epi == EPI_LNMOD) PT_ROW_DISPATCH(EPI_LNMOD)
  if (a.X) PT_ROW_DISPATCH(EPI_ADDLN)
  PT_ROW_DISPATCH(EPI_ADDLN_B)
#undef PT_ROW_DISPATCH
}

}  // namespace pt
