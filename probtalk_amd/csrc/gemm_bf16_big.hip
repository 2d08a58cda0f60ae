// Large-tile bf16 GEMM for the transformer Linears of the bf16 mode: C[M,N] = A[M,K] . W[N,K]^T.
//
// 512 threads = 8 waves (2 x 4), v_mfma_f32_32x32x16_bf16, BK = 64, two LDS stages filled by
// global_load_lds_dwordx4 (LDS-DMA: no staging VGPRs, no ds_write pass).  The LDS image is lane-linear
// per wave-instruction (8 rows x 128 B), so the bank swizzle lives in the per-lane SOURCE address:
// LDS chunk c of row r holds global 16-B chunk c ^ ((r >> 1) & 7), and the fragment reads apply the same
// XOR; every 16-lane group of a ds_read_b128 then hits 16 distinct bank quads.
// MFMA operands are swapped (W fragment first), so a lane owns ONE output row m and 4 consecutive
// columns per register quad: 16-byte fp32 / 8-byte bf16 stores, and row reductions are in-lane.
//
// This is the general kernel: it takes what the phase-pipelined kernels (gemm_bf16_8ph.hip, gemm_bf16_row.hip) do not:
//   256 x 256  (wave tile 128 x 64), EPI_STD: K = 64 (the 8-phase kernel needs two K-tiles in flight)
//   128 x 512  (wave tile 64 x 128), EPI_STD: N = 512 Linears with the fp64 positional table, a residual or zero-padded
//              columns in the epilogue (input / audio / condition embeddings)
//   128 x 512, EPI_LNMOD: out = SiLU(LN(acc + bias) * (1 + scale[label]) + shift[label]) -> bf16 (PTransformer.py:134-139)
//              with more than 4 style classes (the row kernel keeps 4 classes of tables in LDS); a workgroup owns whole
//              rows, the LayerNorm affine is folded into per-class tables G = gamma(1+scale), B = beta(1+scale)+shift
#include <cstdlib>

#include "pt_common.h"

#ifndef PT_DMA_STEPS
#define PT_DMA_STEPS 4
#endif

namespace pt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

void prof_record_begin(hipStream_t s, double flops, int cfg, void** tok);
void prof_record_end(hipStream_t s, void* tok);
bool prof_is_on();

// BM x BN tile, WM x WN waves (wave tile BM/WM x BN/WN), BK = 64 or 32 bf16 per stage.
template <int BM, int BN, int WM, int WN, int BK, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void gemm_big_kernel(const GemmBigArgs p, const int tiles_n, const int ntiles) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int RB = BK * 2;                          // bytes per LDS row
  constexpr int RPD = 1024 / RB;                      // rows per DMA wave-instruction (8 or 16)
  constexpr int CPR = RB / 16;                        // 16-byte chunks per row (8 or 4)
  constexpr int KS = BK / 16;                         // MFMA k-steps per stage
  constexpr int ROWS = BM + BN, NLD = ROWS / (RPD * NW), NLDA = BM / (RPD * NW);
  constexpr int STAGE = ROWS * RB;
  static_assert(ROWS % (RPD * NW) == 0 && BM % (RPD * NW) == 0, "tile rows must split evenly over the DMA instructions");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  // Persistent workgroups: gridDim.x <= #CUs, workgroup b runs tiles order(b), order(b) + G, ...  Inside a round
  // of G tiles the XCD-aware (bijective) remap gives each XCD a contiguous run of tiles (n fastest), so an A
  // row-panel is re-read from one XCD's L2.
  const int G = gridDim.x;
  int tile;
  {
    const int q = G >> 3, rr = G & 7, x = blockIdx.x & 7;
    tile = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (blockIdx.x >> 3);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- staging sources: wave-instruction i of wave w fills tile rows (NW i + w) * RPD .. + RPD-1.  The LDS image
  // is lane-linear (row-major, RB bytes per row); the bank swizzle is applied to the SOURCE chunk index:
  // BK=64: chunk ^= (row>>1)&7, BK=32: chunk ^= (row>>2)&3 (both conflict-free for the ds_read_b128 fragment reads).
  // (32-bit byte offsets from the uniform A / W base: the DMA instructions take base in SGPRs + offset VGPR)
  unsigned soff[NLD];
  auto set_src = [&](int tl) {
    const int tm = tl / tiles_n, tn = tl - tm * tiles_n;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int R = (i * NW + wave) * RPD + lane / CPR;
      const int g = (lane % CPR) ^ (BK == 64 ? ((R >> 1) & 7) : ((R >> 2) & 3));
      if (i < NLDA) {
        int m = tm * BM + R;
        m = m < p.M ? m : p.M - 1;
        soff[i] = ((unsigned)m * (unsigned)p.lda + g * 8) * 2u;
      } else {
        int n = tn * BN + R - BM;
        n = n < p.Npad ? n : p.Npad - 1;
        soff[i] = ((unsigned)n * (unsigned)p.K + g * 8) * 2u;
      }
    }
  };
  // issue wave-instructions [i0, i1) of the stage that fills buffer `buf` with K-tile `kt`
  auto stage = [&](int buf, int kt, int i0, int i1) {
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.A) + (size_t)kt * RB;
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.W) + (size_t)kt * RB;
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (i >= i0 && i < i1)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)((i < NLDA ? abase : wbase) + soff[i]),
                                         (lds_ptr_t)(smem + buf * STAGE + (i * NW + wave) * 1024), 16, 0, 0);
  };

  const int r = lane & 31, h = lane >> 5;
  int off[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
    off[s] = r * RB + (((2 * s + h) ^ (BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3))) << 4);

  // epilogue slabs: wave-private LDS images (32 rows of bf16, or of one fp32 column half) inside the stage buffer the tile used
  // LAST; the other buffer is already receiving the next tile's first K-tile.
  constexpr int PB = TN * 2 + 16, PF = TN * 2 + 16;              // slab row pitches (bf16 rows / fp32 half rows)
  constexpr int SLAB = PB * 32;
  static_assert(NW * SLAB + (EPI == EPI_STD ? 0 : 2 * WN * BM * 4) <= STAGE, "epilogue slabs do not fit one stage buffer");

  const int nt = p.K / BK;
  int gt = 0;                                          // K-tiles done so far (buffer parity runs across tiles)
  if (tile < ntiles) {
    set_src(tile);
    stage(0, 0, 0, NLD);
  }
  for (; tile < ntiles; tile += G) {
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // accumulators start at the bias (its loads overlap the wait for the tile's first K-tile, which is idle time)
    f32x16 acc[MI][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * TN + 4 * (lane >> 5) + ni * 32 + 8 * g;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mi][ni][4 * g + e] = bv[e];
      }

    for (int t = 0; t < nt; ++t, ++gt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                     // K-tile landed for every wave; everyone is done with the other buffer
      const bool more = t + 1 < nt;
      const unsigned char* Ab = smem + (gt & 1) * STAGE + (wm * TM) * RB;
      const unsigned char* Bb = smem + (gt & 1) * STAGE + (BM + wn * TN) * RB;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bf16x8 a[MI], b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(Ab + mi * 32 * RB + off[s]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(Bb + ni * 32 * RB + off[s]);
        // the next K-tile's DMA is spread over the k-steps so its issue hides behind this step's MFMAs
        if (more) {
          constexpr int DS = PT_DMA_STEPS < KS ? PT_DMA_STEPS : KS;     // k-steps that carry DMA issue
          if (s < DS) stage((gt + 1) & 1, t + 1, s * NLD / DS, (s + 1) * NLD / DS);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
      }
    }
    __syncthreads();                                   // every wave is done with the stage buffers of this tile
    unsigned char* ebuf = smem + ((gt - 1) & 1) * STAGE;          // free now: slabs and reduction scratch
    unsigned char* slab = ebuf + wave * SLAB;
    const bool has_next = tile + G < ntiles;

    // ---- epilogue part 1: everything that needs global LOADS, in registers.
    // acc[mi][ni][4g+e]: row m = .. + mi*32 + r, column n = .. + ni*32 + 8g + 4h + e.
    const int mwave = m0 + wm * TM, nwave = n0 + wn * TN;
    const int mrow0 = mwave + r, ncol0 = nwave + 4 * h;
    if constexpr (EPI == EPI_STD) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = ncol0 + ni * 32 + 8 * g;
          const bool real = n < p.N;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int m = mrow0 + mi * 32;
            const bool live = m < p.M;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g + e];
            if (p.pe && real && live) {
              const f64x4 pv = *reinterpret_cast<const f64x4*>(p.pe + (size_t)(m % p.pe_period) * p.N + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (float)((double)v[e] + pv[e]);
            }
            if (p.Res && real && live) v += *reinterpret_cast<const f32x4*>(p.Res + (size_t)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (p.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
              acc[mi][ni][4 * g + e] = real ? v[e] : 0.f;          // columns [N, zero_to) are written as zeros
            }
          }
        }
      }
    } else {
      // whole rows live in this workgroup (BN == N == 512): LayerNorm over the row, two-pass statistics.
      float* red = reinterpret_cast<float*>(ebuf + NW * SLAB);         // [2][WN][BM]
      float mean[MI], rstd[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        float sum = 0.f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
          for (int e = 0; e < 16; ++e) sum += acc[mi][ni][e];
          asm volatile("" : "+v"(acc[mi][ni]) :: "memory");
        }
        sum += __shfl_xor(sum, 32);
        if (h == 0) red[wn * BM + wm * TM + mi * 32 + r] = sum;
      }
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int lr = wm * TM + mi * 32 + r;
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) tot += red[w * BM + lr];
        mean[mi] = tot * (1.f / 512.f);
        float q = 0.f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float d = acc[mi][ni][e] - mean[mi];
            q += d * d;
          }
        q += __shfl_xor(q, 32);
        if (h == 0) red[WN * BM + wn * BM + lr] = q;
      }
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int lr = wm * TM + mi * 32 + r;
        const float* rq = red + WN * BM;
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) tot += rq[w * BM + lr];
        rstd[mi] = __builtin_amdgcn_rsqf(tot * (1.f / 512.f) + 1e-5f);
      }
      // normalise in place: the per-class tables hold gamma*(1+scale) and beta*(1+scale)+shift, so
      // SiLU(LN(v)*(1+scale)+shift) = SiLU((v-mean)*rstd*G[c] + B[c])
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = mrow0 + mi * 32;
        const int mc = m < p.M ? m : p.M - 1;
        long long lb = p.label[p.n_label == 1 ? 0 : mc / p.rows_per_clip];
        lb = lb < 0 ? 0 : (lb >= p.n_classes ? p.n_classes - 1 : lb);
        const float* gp = p.style + (size_t)lb * 1024;
        const float* bp = gp + 512;
        // table rows of the next 32-column block are requested before this block's arithmetic (software pipeline)
        f32x4 ga[2][4], be[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          ga[0][g] = *reinterpret_cast<const f32x4*>(gp + ncol0 + 8 * g);
          be[0][g] = *reinterpret_cast<const f32x4*>(bp + ncol0 + 8 * g);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (ni + 1 < NI) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              ga[(ni + 1) & 1][g] = *reinterpret_cast<const f32x4*>(gp + ncol0 + (ni + 1) * 32 + 8 * g);
              be[(ni + 1) & 1][g] = *reinterpret_cast<const f32x4*>(bp + ncol0 + (ni + 1) * 32 + 8 * g);
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float y = (acc[mi][ni][4 * g + e] - mean[mi]) * rstd[mi] * ga[ni & 1][g][e] + be[ni & 1][g][e];
              y = y * __builtin_amdgcn_rcpf(1.f + __expf(-y));
              acc[mi][ni][4 * g + e] = y;
            }
          }
          asm volatile("" : "+v"(acc[mi][ni]) :: "memory");      // one 32-column block at a time (register pressure)
        }
      }
    }

    // pin the finished values here: without it the compiler sinks the arithmetic below the DMA issue and keeps
    // every epilogue load result live across it (hundreds of spilled registers)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(acc[mi][ni]));

    // ---- the next tile's first K-tile starts flowing into the other buffer while this tile is stored
    if (has_next) {
      set_src(tile + G);
      stage(gt & 1, 0, 0, NLD);
    }

    // ---- epilogue part 2: accumulator layout (a lane owns 4 consecutive columns of ONE row: 32 rows per store
    // instruction) -> wave-private LDS slab -> row-contiguous 16-byte stores.  No global loads from here on.
    const bool want_f32 = EPI == EPI_STD && p.C != nullptr;
    float* outf = p.C;
    const int ldf = p.ldc;
    const int ncols = EPI == EPI_STD ? (p.N > p.zero_to ? p.N : p.zero_to) : 512;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if (want_f32) {
        // fp32: two column halves of the wave tile, 32 rows each (same slab bytes as the bf16 image; no divergent
        // control flow around the LDS traffic).  The bf16 copy is converted on the way out when asked for.
        constexpr int HC = TN / 2, LPR = HC / 4, RPI = 64 / LPR;
        const int row = lane / LPR;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int ni = c * NI / 2; ni < (c + 1) * NI / 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g + e];
              *reinterpret_cast<f32x4*>(slab + r * PF + ((ni - c * NI / 2) * 32 + 8 * g + 4 * h) * 4) = v;
            }
          const int n = nwave + c * HC + (lane % LPR) * 4;
#pragma unroll
          for (int j = 0; j < 32 / RPI; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(slab + (j * RPI + row) * PF + (lane % LPR) * 16);
            const int m = mwave + mi * 32 + j * RPI + row;
            if (m < p.M && n < ncols) {
              *reinterpret_cast<f32x4*>(outf + (size_t)m * ldf + n) = v;
              if (p.Cb) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.Cb) + (size_t)m * p.ldcb + n) = o;
              }
            }
          }
        }
      } else {
        constexpr int LPR = TN / 8, RPI = 64 / LPR;
        const int row = lane / LPR, n = nwave + (lane % LPR) * 8;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[mi][ni][4 * g + e];
            *reinterpret_cast<bf16x4*>(slab + r * PB + (ni * 32 + 8 * g + 4 * h) * 2) = o;
          }
#pragma unroll
        for (int j = 0; j < 32 / RPI; ++j) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(slab + (j * RPI + row) * PB + (lane % LPR) * 16);
          const int m = mwave + mi * 32 + j * RPI + row;
          if (m < p.M && n < ncols) *reinterpret_cast<u32x4*>(p.Cb + (size_t)m * p.ldcb + n) = v;
        }
      }
    }
  }
}

// WPC workgroups per CU are resident (LDS / register budget of the configuration); the persistent grid is WPC x #CUs.
template <int BM, int BN, int WM, int WN, int BK, int EPI, int WPC>
static int launch_big(const GemmBigArgs& a, int cfg, hipStream_t s) {
  constexpr size_t smem = 2 * (size_t)(BM + BN) * BK * 2;
  PT_ENSURE_LDS((gemm_big_kernel<BM, BN, WM, WN, BK, EPI>), smem);
  const int g_num_cus = device_cus();
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
  const int slots = WPC * g_num_cus;
  const int grid = ntiles < slots ? ntiles : slots;      // persistent: resident workgroups loop over the tiles
  void* tok = nullptr;
  if (prof_is_on()) prof_record_begin(s, a.flops > 0 ? a.flops : 2.0 * a.M * a.N * a.K, cfg, &tok);
  PT_LAUNCH((gemm_big_kernel<BM, BN, WM, WN, BK, EPI>), dim3(grid), dim3(64 * WM * WN), smem, s, a, tiles_n, ntiles);
  if (tok) prof_record_end(s, tok);
  return 0;
}

int launch_gemm8(const GemmBigArgs& a, int cfg, hipStream_t s);
int launch_gemm_row(const GemmBigArgs& a, hipStream_t s);

int launch_gemm_big(const GemmBigArgs& a, hipStream_t s) {
  PT_CHECK(a.M > 0 && a.N > 0, "gemm_big: empty problem M=%d N=%d", a.M, a.N);
  PT_CHECK(a.K % 64 == 0 && a.K > 0, "gemm_big: K=%d must be a positive multiple of 64", a.K);
  PT_CHECK(a.lda >= a.K && a.lda % 8 == 0, "gemm_big: lda=%d must be >= K=%d and a multiple of 8", a.lda, a.K);
  PT_CHECK(a.Npad >= a.N, "gemm_big: Npad=%d < N=%d", a.Npad, a.N);
  PT_CHECK(a.N % 8 == 0, "gemm_big: N=%d must be a multiple of 8", a.N);
  PT_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm_big: A and W must be 16-byte aligned");
  PT_CHECK((double)a.M * a.lda * 2 < 4.0e9 && (double)a.Npad * a.K * 2 < 4.0e9, "gemm_big: operand larger than 4 GB");
  const bool row_ok = a.N == 512 && a.Npad == 512;
  if (a.epi == EPI_STD) {
    PT_CHECK(a.C || a.Cb, "gemm_big: no output");
    PT_CHECK((!a.C || (a.ldc >= a.N && a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0)) &&
                 (!a.Cb || (a.ldcb >= a.N && a.ldcb % 8 == 0 && ((uintptr_t)a.Cb & 15) == 0)),
             "gemm_big: output leading dims must be >= N=%d, multiples of 4 (fp32) / 8 (bf16), 16-byte aligned", a.N);
    PT_CHECK(a.zero_to % 8 == 0 && (a.zero_to == 0 || ((!a.C || a.zero_to <= a.ldc) && (!a.Cb || a.zero_to <= a.ldcb))),
             "gemm_big: zero_to=%d must be a multiple of 8 within the output rows", a.zero_to);
    PT_CHECK(!a.Res || (a.ldr % 4 == 0 && ((uintptr_t)a.Res & 15) == 0), "gemm_big: residual must be 16-byte aligned rows");
    PT_CHECK(!a.pe || (a.pe_period > 0 && ((uintptr_t)a.pe & 31) == 0), "gemm_big: pe table needs a period and 32-byte alignment");
    if (gemm_sm_wants(a)) return launch_gemm_sm(a, s);   // latency regime
    if (row_ok && !a.pe && !a.Res && a.zero_to == 0) return launch_gemm_row(a, s);
    if (a.K >= 128 && a.N != 512 && a.N <= 8192) return launch_gemm8(a, 6, s);     // 8192: the bias row it keeps in LDS
    if (a.N == 512) return launch_big<128, 512, 2, 4, 64, EPI_STD, 1>(a, 7, s);
    return launch_big<256, 256, 2, 4, 64, EPI_STD, 1>(a, 6, s);
  }
  PT_CHECK(a.N == 512, "gemm_big: fused LayerNorm epilogues need N = 512 (got %d)", a.N);
  PT_CHECK(a.bias && a.Cb && a.ldcb >= 512 && a.ldcb % 8 == 0 && ((uintptr_t)a.Cb & 15) == 0,
           "gemm_big: fused LayerNorm epilogue needs a bias and a bf16 output");
  if (a.epi == EPI_LNMOD) {
    PT_CHECK(a.style && a.label && a.n_label >= 1 && a.rows_per_clip > 0 && a.n_classes > 0,
             "gemm_big: LNMOD epilogue needs the style table and labels");
    PT_CHECK(a.n_label == 1 || (long long)a.n_label * a.rows_per_clip >= a.M, "gemm_big: %d labels x %d rows < M=%d",
             a.n_label, a.rows_per_clip, a.M);
    if (row_ok && a.n_classes <= 4) return launch_gemm_row(a, s);      // the row kernel keeps 4 classes of style tables in LDS
    return launch_big<128, 512, 2, 4, 64, EPI_LNMOD, 1>(a, 7, s);
  }
  PT_CHECK(a.epi == EPI_ADDLN && a.gamma && a.beta && ((uintptr_t)a.X & 15) == 0,
           "gemm_big: ADDLN epilogue needs gamma, beta and a 16-byte aligned residual stream");
  PT_CHECK(row_ok, "gemm_big: the residual + LayerNorm epilogue needs N = Npad = 512");
  return launch_gemm_row(a, s);          // X == nullptr: bf16 residual stream, Cb updated in place
}

}  // namespace pt
