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
// Two tile shapes:
//   256 x 256  (wave tile 128 x 64): in-proj, FFN-1, heads       -> EPI_STD epilogue
//   128 x 512  (wave tile 64 x 128): every N = 512 Linear; a workgroup owns whole rows, so the
//       LayerNorm that follows the Linear in the decoder layer is fused into the epilogue:
//       EPI_LNMOD  out = SiLU(LN(acc + bias) * (1 + scale[label]) + shift[label])  -> bf16   (PTransformer.py:134-139)
//       EPI_ADDLN  x   = LN(x + acc + bias)                                       -> fp32 x and bf16 copy (:329-331)
#include "pt_common.h"

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

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(512) void gemm_big_kernel(const GemmBigArgs p, const int tiles_n, const int nwg) {
  constexpr int WN = 4;
  constexpr int TM = BM / 2, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int ROWS = BM + BN, NLD = ROWS / 64;
  constexpr int STAGE = ROWS * 128;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- staging sources: wave-instruction i of wave w fills tile rows (8 i + w) * 8 .. + 7
  const unsigned short* src[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int R = (i * 8 + wave) * 8 + (lane >> 3);
    const int g = (lane & 7) ^ ((R >> 1) & 7);
    if (R < BM) {
      int m = m0 + R;
      m = m < p.M ? m : p.M - 1;
      src[i] = p.A + (size_t)m * p.lda + g * 8;
    } else {
      int n = n0 + R - BM;
      n = n < p.Npad ? n : p.Npad - 1;
      src[i] = p.W + (size_t)n * p.K + g * 8;
    }
  }
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[i] + kt * 64),
                                       (lds_ptr_t)(smem + buf * STAGE + (i * 8 + wave) * 1024), 16, 0, 0);
  };

  const int r = lane & 31, h = lane >> 5;
  int off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) off[s] = r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) << 4);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int nt = p.K / 64;
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // tile t landed for every wave; everyone is done with the other buffer
    if (t + 1 < nt) stage((t + 1) & 1, t + 1);
    const unsigned char* Ab = smem + (t & 1) * STAGE + (wm * TM) * 128;
    const unsigned char* Bb = smem + (t & 1) * STAGE + (BM + wn * TN) * 128;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(Ab + mi * 4096 + off[s]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(Bb + ni * 4096 + off[s]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
    }
  }

  // ---- epilogue.  acc[mi][ni][4g+e]: row m = .. + mi*32 + r, column n = .. + ni*32 + 8g + 4h + e.
  // Outputs leave through a wave-private LDS slab (32 rows x TN columns) that turns the accumulator layout
  // (a lane owns 4 consecutive columns of ONE row: 32 rows per store instruction) into row-contiguous 16-byte
  // stores (a store instruction covers whole 128..512-byte row segments).
  constexpr int SLAB = (TN * 4 + 16) * 32;                       // fp32 slab, the larger of the two
  static_assert(8 * SLAB + 4096 <= 2 * STAGE, "epilogue slabs do not fit the stage memory");
  unsigned char* slab = smem + wave * SLAB;
  const int mwave = m0 + wm * TM, nwave = n0 + wn * TN;
  constexpr int PB = TN * 2 + 16, PF = TN * 4 + 16;              // slab row pitches (bf16 / fp32)
  auto put_bf16 = [&](int ni, int g, const f32x4& v) {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x4*>(slab + r * PB + (ni * 32 + 8 * g + 4 * h) * 2) = o;
  };
  auto put_f32 = [&](int ni, int g, const f32x4& v) {
    *reinterpret_cast<f32x4*>(slab + r * PF + (ni * 32 + 8 * g + 4 * h) * 4) = v;
  };
  // slab rows -> global, 16 bytes per lane along the row
  auto flush_bf16 = [&](int mi, unsigned short* out, int ld, int ncols) {
    constexpr int LPR = TN / 8, RPI = 64 / LPR;
    const int row = lane / LPR, n = nwave + (lane % LPR) * 8;
#pragma unroll
    for (int j = 0; j < 32 / RPI; ++j) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(slab + (j * RPI + row) * PB + (lane % LPR) * 16);
      const int m = mwave + mi * 32 + j * RPI + row;
      if (m < p.M && n < ncols) *reinterpret_cast<u32x4*>(out + (size_t)m * ld + n) = v;
    }
  };
  auto flush_f32 = [&](int mi, float* outf, int ldf, unsigned short* outb, int ldb, int ncols) {
    constexpr int LPR = TN / 4, RPI = 64 / LPR;
    const int row = lane / LPR, n = nwave + (lane % LPR) * 4;
#pragma unroll
    for (int j = 0; j < 32 / RPI; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(slab + (j * RPI + row) * PF + (lane % LPR) * 16);
      const int m = mwave + mi * 32 + j * RPI + row;
      if (m < p.M && n < ncols) {
        if (outf) *reinterpret_cast<f32x4*>(outf + (size_t)m * ldf + n) = v;
        if (outb) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
          *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(outb) + (size_t)m * ldb + n) = o;
        }
      }
    }
  };

  const int mrow0 = mwave + r, ncol0 = nwave + 4 * h;
  __syncthreads();                                  // every wave is done with the stage buffers
  if constexpr (EPI == EPI_STD) {
    const int ncols = p.N > p.zero_to ? p.N : p.zero_to;   // columns [N, zero_to) are written as zeros
    const bool via_f32 = p.C != nullptr;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = mrow0 + mi * 32;
      const bool live = m < p.M;
      const double* pe_row = (p.pe && live) ? p.pe + (size_t)(m % p.pe_period) * p.N : nullptr;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = ncol0 + ni * 32 + 8 * g;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (n < p.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g + e];
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (pe_row) {
              const f64x4 pv = *reinterpret_cast<const f64x4*>(pe_row + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (float)((double)v[e] + pv[e]);
            }
            if (p.Res && live) v += *reinterpret_cast<const f32x4*>(p.Res + (size_t)m * p.ldr + n);
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
          }
          if (via_f32) put_f32(ni, g, v); else put_bf16(ni, g, v);
        }
      }
      if (via_f32) flush_f32(mi, p.C, p.ldc, p.Cb, p.ldcb, ncols); else flush_bf16(mi, p.Cb, p.ldcb, ncols);
    }
  } else {
    // whole rows live in this workgroup (BN == N == 512): LayerNorm over the row, two-pass statistics.
    float* red = reinterpret_cast<float*>(smem + 8 * SLAB);          // [2][WN][BM]
    float mean[MI], rstd[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = mrow0 + mi * 32;
      const bool live = m < p.M;
      float s = 0.f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = ncol0 + ni * 32 + 8 * g;
          f32x4 add = *reinterpret_cast<const f32x4*>(p.bias + n);
          if (EPI == EPI_ADDLN && live) add += *reinterpret_cast<const f32x4*>(p.X + (size_t)m * 512 + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[mi][ni][4 * g + e] += add[e];
            s += acc[mi][ni][4 * g + e];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      s += __shfl_xor(s, 32);
      if (h == 0) red[wn * BM + wm * TM + mi * 32 + r] = s;
    }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int lr = wm * TM + mi * 32 + r;
      mean[mi] = (red[lr] + red[BM + lr] + red[2 * BM + lr] + red[3 * BM + lr]) * (1.f / 512.f);
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
      rstd[mi] = 1.0f / sqrtf((rq[lr] + rq[BM + lr] + rq[2 * BM + lr] + rq[3 * BM + lr]) * (1.f / 512.f) + 1e-5f);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = mrow0 + mi * 32;
      const float* st = nullptr;
      if (EPI == EPI_LNMOD) {
        const int mc = m < p.M ? m : p.M - 1;
        long long lb = p.label[p.n_label == 1 ? 0 : mc / p.rows_per_clip];
        lb = lb < 0 ? 0 : (lb >= p.n_classes ? p.n_classes - 1 : lb);
        st = p.style + (size_t)lb * 1024;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = ncol0 + ni * 32 + 8 * g;
          const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + n);
          const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + n);
          f32x4 y;
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (acc[mi][ni][4 * g + e] - mean[mi]) * rstd[mi] * ga[e] + be[e];
          if (EPI == EPI_LNMOD) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(st + n);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(st + 512 + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = y[e] * (1.f + sc[e]) + sh[e];
              y[e] = t / (1.f + expf(-t));
            }
            put_bf16(ni, g, y);
          } else {
            put_f32(ni, g, y);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (EPI == EPI_ADDLN) flush_f32(mi, p.X, 512, p.Cb, p.ldcb, 512);
      else flush_bf16(mi, p.Cb, p.ldcb, 512);
    }
  }
}

template <int BM, int BN, int EPI>
static int launch_big(const GemmBigArgs& a, int cfg, hipStream_t s) {
  constexpr size_t smem = 2 * (size_t)(BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    PT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_kernel<BM, BN, EPI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, nwg = tiles_m * tiles_n;
  void* tok = nullptr;
  if (prof_is_on()) prof_record_begin(s, a.flops > 0 ? a.flops : 2.0 * a.M * a.N * a.K, cfg, &tok);
  PT_LAUNCH((gemm_big_kernel<BM, BN, EPI>), dim3(nwg), dim3(512), smem, s, a, tiles_n, nwg);
  if (tok) prof_record_end(s, tok);
  return 0;
}

int launch_gemm_big(const GemmBigArgs& a, hipStream_t s) {
  PT_CHECK(a.M > 0 && a.N > 0, "gemm_big: empty problem M=%d N=%d", a.M, a.N);
  PT_CHECK(a.K % 64 == 0 && a.K > 0, "gemm_big: K=%d must be a positive multiple of 64", a.K);
  PT_CHECK(a.lda >= a.K && a.lda % 8 == 0, "gemm_big: lda=%d must be >= K=%d and a multiple of 8", a.lda, a.K);
  PT_CHECK(a.Npad >= a.N, "gemm_big: Npad=%d < N=%d", a.Npad, a.N);
  PT_CHECK(a.N % 8 == 0, "gemm_big: N=%d must be a multiple of 8", a.N);
  PT_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm_big: A and W must be 16-byte aligned");
  if (a.epi == EPI_STD) {
    PT_CHECK(a.C || a.Cb, "gemm_big: no output");
    PT_CHECK((!a.C || (a.ldc >= a.N && a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0)) &&
                 (!a.Cb || (a.ldcb >= a.N && a.ldcb % 8 == 0 && ((uintptr_t)a.Cb & 15) == 0)),
             "gemm_big: output leading dims must be >= N=%d, multiples of 4 (fp32) / 8 (bf16), 16-byte aligned", a.N);
    PT_CHECK(a.zero_to % 8 == 0 && (a.zero_to == 0 || ((!a.C || a.zero_to <= a.ldc) && (!a.Cb || a.zero_to <= a.ldcb))),
             "gemm_big: zero_to=%d must be a multiple of 8 within the output rows", a.zero_to);
    PT_CHECK(!a.Res || (a.ldr % 4 == 0 && ((uintptr_t)a.Res & 15) == 0), "gemm_big: residual must be 16-byte aligned rows");
    PT_CHECK(!a.pe || (a.pe_period > 0 && ((uintptr_t)a.pe & 31) == 0), "gemm_big: pe table needs a period and 32-byte alignment");
    if (a.N == 512) return launch_big<128, 512, EPI_STD>(a, 7, s);
    return launch_big<256, 256, EPI_STD>(a, 6, s);
  }
  PT_CHECK(a.N == 512, "gemm_big: fused LayerNorm epilogues need N = 512 (got %d)", a.N);
  PT_CHECK(a.bias && a.gamma && a.beta && a.Cb && a.ldcb >= 512 && a.ldcb % 8 == 0 && ((uintptr_t)a.Cb & 15) == 0,
           "gemm_big: fused LayerNorm epilogue needs bias, gamma, beta and a bf16 output");
  if (a.epi == EPI_LNMOD) {
    PT_CHECK(a.style && a.label && a.n_label >= 1 && a.rows_per_clip > 0 && a.n_classes > 0,
             "gemm_big: LNMOD epilogue needs the style table and labels");
    PT_CHECK(a.n_label == 1 || (long long)a.n_label * a.rows_per_clip >= a.M, "gemm_big: %d labels x %d rows < M=%d",
             a.n_label, a.rows_per_clip, a.M);
    return launch_big<128, 512, EPI_LNMOD>(a, 7, s);
  }
  PT_CHECK(a.epi == EPI_ADDLN && a.X && ((uintptr_t)a.X & 15) == 0, "gemm_big: ADDLN epilogue needs the residual stream X");
  return launch_big<128, 512, EPI_ADDLN>(a, 7, s);
}

}  // namespace pt
