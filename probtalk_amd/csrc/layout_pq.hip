// Layout changes at the library boundary (reference tensors are NCL, the library computes
// channels-last) and the product quantizer: nearest-code search on the fp32 matrix cores with a
// fused wavefront argmin (lowest index wins ties, like torch.argmin) and the code gather.
#include "pt_common.h"

namespace pt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// (B, C, L) -> rows (b*L + l) of ldo floats at column col_off; columns [C, zero_to-col_off) zeroed.
// ---------------------------------------------------------------------------
template <typename OutT>
__global__ __launch_bounds__(256) void ncl_to_cl_kernel(const float* __restrict__ in, OutT* __restrict__ out, int C,
                                                        int L, int ldo, int col_off, int ccover) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, l0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, l = l0 + tx;
    tile[ty + 8 * i][tx] = (c < C && l < L) ? in[((size_t)b * C + c) * L + l] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int l = l0 + ty + 8 * i, c = c0 + tx;
    if (l < L && c < ccover) out[((size_t)b * L + l) * ldo + col_off + c] = (OutT)tile[tx][ty + 8 * i];
  }
}

int launch_ncl_to_cl(const float* in, float* out, int B, int C, int L, int ldo, int col_off, int zero_to,
                     hipStream_t s) {
  PT_CHECK(B > 0 && C > 0 && L > 0, "ncl_to_cl: empty tensor B=%d C=%d L=%d", B, C, L);
  const int ccover = (zero_to > col_off + C ? zero_to - col_off : C);
  PT_CHECK(col_off + ccover <= ldo, "ncl_to_cl: columns %d..%d exceed ld %d", col_off, col_off + ccover, ldo);
  dim3 grid((L + 31) / 32, (ccover + 31) / 32, B);
  PT_LAUNCH(ncl_to_cl_kernel<float>, grid, dim3(256), 0, s, in, out, C, L, ldo, col_off, ccover);
  return 0;
}
// same with bf16 rows (bf16 mode of the conv stacks)
int launch_ncl_to_cl_b(const float* in, unsigned short* out, int B, int C, int L, int ldo, int col_off, int zero_to,
                       hipStream_t s) {
  PT_CHECK(B > 0 && C > 0 && L > 0, "ncl_to_cl: empty tensor B=%d C=%d L=%d", B, C, L);
  const int ccover = (zero_to > col_off + C ? zero_to - col_off : C);
  PT_CHECK(col_off + ccover <= ldo, "ncl_to_cl: columns %d..%d exceed ld %d", col_off, col_off + ccover, ldo);
  dim3 grid((L + 31) / 32, (ccover + 31) / 32, B);
  PT_LAUNCH(ncl_to_cl_kernel<__bf16>, grid, dim3(256), 0, s, in, reinterpret_cast<__bf16*>(out), C, L, ldo, col_off, ccover);
  return 0;
}

__global__ __launch_bounds__(256) void cl_to_ncl_kernel(const float* __restrict__ in, int ldi, int col_off,
                                                        float* __restrict__ out, int C, int L) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, l0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int l = l0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (c < C && l < L) ? in[((size_t)b * L + l) * ldi + col_off + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, l = l0 + tx;
    if (c < C && l < L) out[((size_t)b * C + c) * L + l] = tile[tx][ty + 8 * i];
  }
}

int launch_cl_to_ncl(const float* in, int ldi, int col_off, float* out, int B, int C, int L, hipStream_t s) {
  PT_CHECK(B > 0 && C > 0 && L > 0, "cl_to_ncl: empty tensor B=%d C=%d L=%d", B, C, L);
  dim3 grid((L + 31) / 32, (C + 31) / 32, B);
  PT_LAUNCH(cl_to_ncl_kernel, grid, dim3(256), 0, s, in, ldi, col_off, out, C, L);
  return 0;
}

// ---------------------------------------------------------------------------
// PQ search.  One workgroup = 64 latent rows of one sub-codebook g; codes are swept in tiles of 64.
// d[row][code] = (|x|^2 + |e|^2) - 2 x.e   in the reference's association order
// (vqvae_modules.py:549-553); x.e on v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain).
// 4 waves = 2 (row halves) x 2 (code halves).  The code fragment is the MFMA's first operand, so a lane's 16 result
// registers are 16 CODES of one latent row (the lane index): the lane keeps a running (min, argmin) for its row over the
// codes it sees (ascending, strict <), then one exchange with the lane holding the other 16 codes of each tile and a
// 2-way LDS merge, both breaking ties towards the lower code index.  The winning code rows are gathered into q.
// ---------------------------------------------------------------------------
constexpr int PQ_ROWS = 64, PQ_CODES = 64;

__device__ __forceinline__ void argmin_merge(float& bd, int& bi, float od, int oi) {
  if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
}

__global__ __launch_bounds__(256) void pq_search_kernel(const float* __restrict__ z, int ldz,
                                                        const float* __restrict__ codebooks, int rows, int G, int D,
                                                        int K, int64_t* __restrict__ idx, float* __restrict__ q,
                                                        int ldq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int st = D + 4;
  float* Xs = smem;                       // [64][st]
  float* Es = Xs + PQ_ROWS * st;          // [64][st]
  float* xxs = Es + PQ_CODES * st;        // [64]
  float* ees = xxs + PQ_ROWS;             // [64]
  float* cand_d = ees + PQ_CODES;         // [2][64]
  int* cand_i = reinterpret_cast<int*>(cand_d + 2 * PQ_ROWS);  // [2][64]
  int* win = cand_i + 2 * PQ_ROWS;        // [64]

  const int g = blockIdx.y, row0 = blockIdx.x * PQ_ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, h = lane >> 5, l31 = lane & 31;
  const int d4 = D / 4;
  const float* E = codebooks + (size_t)g * K * D;

  for (int e = tid; e < PQ_ROWS * d4; e += 256) {
    const int r = e / d4, c = e - r * d4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + r < rows) v = *reinterpret_cast<const f32x4*>(z + (size_t)(row0 + r) * ldz + g * D + c * 4);
    *reinterpret_cast<f32x4*>(Xs + r * st + c * 4) = v;
  }
  __syncthreads();
  {  // |x|^2 per row: 4 threads per row
    const int r = tid >> 2, part = tid & 3, n = D / 4;
    float s = 0.f;
    for (int k = part * n; k < (part + 1) * n; ++k) s += Xs[r * st + k] * Xs[r * st + k];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (part == 0) xxs[r] = s;
  }

  float best_d = INFINITY;
  int best_i = 0x7fffffff;

  for (int c0 = 0; c0 < K; c0 += PQ_CODES) {
    __syncthreads();
    for (int e = tid; e < PQ_CODES * d4; e += 256) {
      const int r = e / d4, c = e - r * d4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c0 + r < K) v = *reinterpret_cast<const f32x4*>(E + (size_t)(c0 + r) * D + c * 4);
      *reinterpret_cast<f32x4*>(Es + r * st + c * 4) = v;
    }
    __syncthreads();
    {
      const int r = tid >> 2, part = tid & 3, n = D / 4;
      float s = 0.f;
      for (int k = part * n; k < (part + 1) * n; ++k) s += Es[r * st + k] * Es[r * st + k];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (part == 0) ees[r] = (c0 + r < K) ? s : INFINITY;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* xa = Xs + (wr * 32 + l31) * st + 4 * h;
    const float* eb = Es + (wc * 32 + l31) * st + 4 * h;
    for (int u = 0; u < D / 8; ++u) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xa + u * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(eb + u * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[j], acc, 0, 0, 0);
    }
    const float xx = xxs[wr * 32 + l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                // ascending codes within the lane
      const int cl = wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float d = (xx + ees[cl]) - 2.f * acc[r];
      if (d < best_d) { best_d = d; best_i = c0 + cl; }
    }
  }
  {
    const float od = __shfl_xor(best_d, 32);
    const int oi = __shfl_xor(best_i, 32);
    argmin_merge(best_d, best_i, od, oi);
    if (h == 0) {
      cand_d[wc * PQ_ROWS + wr * 32 + l31] = best_d;
      cand_i[wc * PQ_ROWS + wr * 32 + l31] = best_i;
    }
  }
  __syncthreads();
  if (tid < PQ_ROWS) {
    float bd = cand_d[tid];
    int bi = cand_i[tid];
    argmin_merge(bd, bi, cand_d[PQ_ROWS + tid], cand_i[PQ_ROWS + tid]);
    if (bi < 0 || bi >= K) bi = 0;  // all-NaN row: keep memory accesses in range
    win[tid] = bi;
    if (row0 + tid < rows && idx) idx[(size_t)(row0 + tid) * G + g] = bi;
  }
  __syncthreads();
  if (q) {
    for (int e = tid; e < PQ_ROWS * d4; e += 256) {
      const int r = e / d4, c = e - r * d4;
      if (row0 + r < rows)
        *reinterpret_cast<f32x4*>(q + (size_t)(row0 + r) * ldq + g * D + c * 4) =
            *reinterpret_cast<const f32x4*>(E + (size_t)win[r] * D + c * 4);
    }
  }
}

// ---------------------------------------------------------------------------
// PQ search, sub-codebook RESIDENT in LDS (round 3).  The kernel above re-stages its 64-code tiles and their norms for every
// 64 latent rows: a workgroup lives 34 us for 4 us of MFMA work (bench.py --workload vqvae: 0.98 TB/s of algorithmic
// bytes, 31 TFLOP/s).  Here a workgroup keeps ONE whole sub-codebook (K x D floats + |e|^2: 68 KB at 128 x 128) and walks
// the 64-row tiles of the batch: per tile only the latent rows come in (register prefetch of the next tile under the
// MFMAs of the current one, two LDS buffers), the winning code rows are gathered from LDS, not from memory.
// 8 waves = 2 (row halves) x 4 (code tiles of 32, strided over K); same arithmetic as above to the bit: the same k order
// in the MFMA chain, the same (|x|^2 + |e|^2) - 2 x.e association, the same 4-partial norms, strict < over ascending
// codes and lowest-index merges (tests/test_gpu_vqvae.py compares the two kernels bit for bit).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void pq_search_res_kernel(const float* __restrict__ z, int ldz,
                                                            const float* __restrict__ codebooks, int rows, int G, int D,
                                                            int K, int64_t* __restrict__ idx, float* __restrict__ q,
                                                            int ldq, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int st = D + 4;
  float* Es = smem;                           // [K][st]
  float* Xs = Es + (size_t)K * st;            // [2][64][st]
  float* ees = Xs + 2 * PQ_ROWS * st;         // [K]
  float* xxs = ees + K;                       // [64]
  float* cand_d = xxs + PQ_ROWS;              // [4][64]
  int* cand_i = reinterpret_cast<int*>(cand_d + 4 * PQ_ROWS);   // [4][64]
  int* win = cand_i + 4 * PQ_ROWS;            // [64]

  const int g = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 1, wc = wave >> 1, h = lane >> 5, l31 = lane & 31;
  const int d4 = D / 4, per = PQ_ROWS * d4;   // float4 per latent tile
  const float* E = codebooks + (size_t)g * K * D;

  for (int e = tid; e < K * d4; e += 512) {
    const int r = e / d4, c = e - r * d4;
    *reinterpret_cast<f32x4*>(Es + (size_t)r * st + c * 4) = *reinterpret_cast<const f32x4*>(E + (size_t)r * D + c * 4);
  }
  __syncthreads();
  for (int e = tid; e < K * 4; e += 512) {    // |e|^2: 4 partial sums per code, combined (p0 + p1) + (p2 + p3)
    const int r = e >> 2, part = e & 3, n = D / 4;
    float s = 0.f;
    for (int k = part * n; k < (part + 1) * n; ++k) s += Es[(size_t)r * st + k] * Es[(size_t)r * st + k];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (part == 0) ees[r] = s;
  }

  // this thread's share of a latent tile: float4 e = tid + 512 i (D <= 256: at most 8)
  f32x4 xr[8];
  auto gload = [&](int tile) {
    const int row0 = tile * PQ_ROWS;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 512 * i;
      if (e < per) {
        const int r = e / d4, c = e - r * d4;
        xr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row0 + r < rows) xr[i] = *reinterpret_cast<const f32x4*>(z + (size_t)(row0 + r) * ldz + g * D + c * 4);
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 512 * i;
      if (e < per) {
        const int r = e / d4, c = e - r * d4;
        *reinterpret_cast<f32x4*>(Xs + (size_t)buf * PQ_ROWS * st + r * st + c * 4) = xr[i];
      }
    }
  };
  int tile = blockIdx.x, buf = 0;
  if (tile < ntiles) {
    gload(tile);
    lstore(0);
  }
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const float* X = Xs + (size_t)buf * PQ_ROWS * st;
    const int row0 = tile * PQ_ROWS;
    const bool more = tile + (int)gridDim.x < ntiles;
    __syncthreads();                                            // X(buf) and the code norms are in LDS; win / cand are free again
    if (more) gload(tile + gridDim.x);
    if (tid < 256) {  // |x|^2 per row: 4 threads per row
      const int r = tid >> 2, part = tid & 3, n = D / 4;
      float s = 0.f;
      for (int k = part * n; k < (part + 1) * n; ++k) s += X[r * st + k] * X[r * st + k];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (part == 0) xxs[r] = s;
    }
    __syncthreads();
    // MFMA operands swapped against the staging kernel (code fragment first): the 32 x 32 result then has the CODES in a
    // lane's 16 registers and the latent ROW in the lane index, so the running argmin of a row is register work, one
    // lane exchange (the two 16-code halves of a tile sit in lanes l and l + 32) and the merge over the four code-tile
    // waves, instead of a 32-lane butterfly per row (160 LDS-crossbar permutes per wave and tile).  Same products, same
    // k order, same (|x|^2 + |e|^2) - 2 x.e: the same bits.
    float best_d = INFINITY;
    int best_i = 0x7fffffff;
    const float* xa = X + (wr * 32 + l31) * st + 4 * h;
    const float xx = xxs[wr * 32 + l31];
    for (int c0 = wc * 32; c0 < K; c0 += 128) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int crow = c0 + l31 < K ? c0 + l31 : K - 1;
      const float* eb = Es + (size_t)crow * st + 4 * h;
      for (int u = 0; u < D / 8; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xa + u * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(eb + u * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[j], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {                              // ascending codes within the lane
        const int code = c0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float e2 = code < K ? ees[code] : INFINITY;
        const float d = (xx + e2) - 2.f * acc[r];
        if (d < best_d) { best_d = d; best_i = code; }
      }
    }
    {
      const float od = __shfl_xor(best_d, 32);
      const int oi = __shfl_xor(best_i, 32);
      argmin_merge(best_d, best_i, od, oi);
      if (h == 0) {
        cand_d[wc * PQ_ROWS + wr * 32 + l31] = best_d;
        cand_i[wc * PQ_ROWS + wr * 32 + l31] = best_i;
      }
    }
    __syncthreads();
    if (tid < PQ_ROWS) {
      float bd = cand_d[tid];
      int bi = cand_i[tid];
#pragma unroll
      for (int w = 1; w < 4; ++w) argmin_merge(bd, bi, cand_d[w * PQ_ROWS + tid], cand_i[w * PQ_ROWS + tid]);
      if (bi < 0 || bi >= K) bi = 0;  // all-NaN row: keep memory accesses in range
      win[tid] = bi;
      if (row0 + tid < rows && idx) idx[(size_t)(row0 + tid) * G + g] = bi;
    }
    if (more) lstore(buf ^ 1);                                  // the other buffer was last read one tile ago (two barriers back)
    __syncthreads();
    if (q) {
      for (int e = tid; e < per; e += 512) {
        const int r = e / d4, c = e - r * d4;
        if (row0 + r < rows)
          *reinterpret_cast<f32x4*>(q + (size_t)(row0 + r) * ldq + g * D + c * 4) =
              *reinterpret_cast<const f32x4*>(Es + (size_t)win[r] * st + c * 4);
      }
    }
  }
}
static size_t pq_search_res_smem(int D, int K) {
  return ((size_t)(K + 2 * PQ_ROWS) * (D + 4) + K + PQ_ROWS + 9 * PQ_ROWS) * sizeof(float);
}
static LdsAttr pq_res_lds_attr;
// the resident form applies when the sub-codebook and two latent tiles fit the CU's LDS and the batch gives every
// workgroup a few tiles to walk (below that the staging kernel's finer grid wins)
static bool pq_search_resident(int rows, int G, int D, int K) {
  return pq_search_res_smem(D, K) <= 160 * 1024 && (long long)((rows + PQ_ROWS - 1) / PQ_ROWS) * G >= 2LL * device_cus();
}

static size_t pq_search_smem(int D) { return ((size_t)(PQ_ROWS + PQ_CODES) * (D + 4) + PQ_ROWS + PQ_CODES + 5 * PQ_ROWS) * sizeof(float); }
static LdsAttr pq_lds_attr;

int pq_search_prepare(int D, int K) {
  PT_CHECK(D % 8 == 0 && D >= 8 && D <= 256, "pq_search: code dim %d must be a multiple of 8 in [8,256]", D);
  if (pq_search_res_smem(D, K) <= 160 * 1024 &&
      ensure_dyn_lds(reinterpret_cast<const void*>(pq_search_res_kernel), (int)pq_search_res_smem(D, K), pq_res_lds_attr)) return 1;
  return ensure_dyn_lds(reinterpret_cast<const void*>(pq_search_kernel), (int)pq_search_smem(D), pq_lds_attr);
}

int launch_pq_search(const float* z_cl, int ldz, const float* codebooks, int rows, int G, int D, int K, int64_t* idx,
                     float* q_cl, int ldq, hipStream_t s, bool prepared) {
  PT_CHECK(rows > 0 && G > 0 && K > 0, "pq_search: empty problem rows=%d G=%d K=%d", rows, G, K);
  PT_CHECK(D % 8 == 0 && D >= 8 && D <= 256, "pq_search: code dim %d must be a multiple of 8 in [8,256]", D);
  PT_CHECK(ldz >= G * D && ldz % 4 == 0 && (q_cl == nullptr || (ldq >= G * D && ldq % 4 == 0)),
           "pq_search: leading dims ldz=%d ldq=%d too small for G*D=%d", ldz, ldq, G * D);
  if (pq_search_resident(rows, G, D, K)) {
    const size_t rs = pq_search_res_smem(D, K);
    if (!prepared && ensure_dyn_lds(reinterpret_cast<const void*>(pq_search_res_kernel), (int)rs, pq_res_lds_attr)) return 1;
    const int ntiles = (rows + PQ_ROWS - 1) / PQ_ROWS;
    int per_group = device_cus() / G;                             // one workgroup per CU over all groups
    per_group = per_group < 1 ? 1 : (per_group > ntiles ? ntiles : per_group);
    PT_LAUNCH(pq_search_res_kernel, dim3(per_group, G), dim3(512), rs, s, z_cl, ldz, codebooks, rows, G, D, K, idx, q_cl, ldq, ntiles);
    return 0;
  }
  const size_t smem = pq_search_smem(D);
  if (!prepared && ensure_dyn_lds(reinterpret_cast<const void*>(pq_search_kernel), (int)smem, pq_lds_attr)) return 1;
  dim3 grid((rows + PQ_ROWS - 1) / PQ_ROWS, G);
  PT_LAUNCH(pq_search_kernel, grid, dim3(256), smem, s, z_cl, ldz, codebooks, rows, G, D, K, idx, q_cl, ldq);
  return 0;
}

// quantize_all (vqvae_modules.py:557-564): q[row][g*D + :] = E_g[idx[row][g]]
__global__ __launch_bounds__(256) void pq_gather_kernel(const void* __restrict__ idx, int is64,
                                                        const float* __restrict__ codebooks, int rows, int G, int D,
                                                        int K, float* __restrict__ q, int ldq, int zero_to) {
  const int d4 = D / 4, per_row = G * d4, zc = (zero_to - G * D) / 4;
  const size_t total = (size_t)rows * (per_row + (zc > 0 ? zc : 0));
  const int stride_row = per_row + (zc > 0 ? zc : 0);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int row = (int)(e / stride_row), c = (int)(e - (size_t)row * stride_row);
    if (c >= per_row) {
      *reinterpret_cast<f32x4*>(q + (size_t)row * ldq + G * D + (c - per_row) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const int g = c / d4, cc = c - g * d4;
    long long i = is64 ? reinterpret_cast<const long long*>(idx)[(size_t)row * G + g]
                       : (long long)reinterpret_cast<const int*>(idx)[(size_t)row * G + g];
    if (i < 0) i = 0;
    if (i >= K) i = K - 1;
    *reinterpret_cast<f32x4*>(q + (size_t)row * ldq + g * D + cc * 4) =
        *reinterpret_cast<const f32x4*>(codebooks + ((size_t)g * K + i) * D + cc * 4);
  }
}

int launch_pq_gather(const void* idx, int idx_is_i64, const float* codebooks, int rows, int G, int D, int K,
                     float* q_cl, int ldq, int zero_to, hipStream_t s) {
  PT_CHECK(rows > 0 && G > 0 && K > 0 && D % 4 == 0, "pq_gather: bad problem rows=%d G=%d D=%d K=%d", rows, G, D, K);
  PT_CHECK(ldq >= G * D && ldq % 4 == 0 && zero_to <= ldq, "pq_gather: ldq=%d too small", ldq);
  const size_t total = (size_t)rows * (ldq / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  PT_LAUNCH(pq_gather_kernel, dim3(blocks), dim3(256), 0, s, idx, idx_is_i64, codebooks, rows, G, D, K, q_cl,
                     ldq, zero_to);
  return 0;
}

}  // namespace pt
