// Row-wise (HBM-bound) kernels of the predictor / refiner transformers: token embedding + fp64
// positional encodings, the identity-conditioned "stylization" pre-op (LayerNorm -> (1+scale),shift
// -> SiLU), post-LN residual, the MaskGIT re-masking step and linear resize + mask blend.
// d_model is 512 on this path: one 64-lane wave owns one row = 8 floats per lane (2 x dwordx4).
#include <algorithm>
#include "transformer.h"

namespace pt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DM = 512;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

struct Row8 {
  f32x4 a, b;
};
__device__ __forceinline__ Row8 load_row(const float* p, int lane) {
  Row8 r;
  r.a = *reinterpret_cast<const f32x4*>(p + lane * 4);
  r.b = *reinterpret_cast<const f32x4*>(p + 256 + lane * 4);
  return r;
}
__device__ __forceinline__ void store_row(float* p, int lane, const Row8& r) {
  *reinterpret_cast<f32x4*>(p + lane * 4) = r.a;
  *reinterpret_cast<f32x4*>(p + 256 + lane * 4) = r.b;
}
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_row_bf16(unsigned short* p, int lane, const Row8& r) {
  bf16x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = (__bf16)r.a[i]; b[i] = (__bf16)r.b[i]; }
  *reinterpret_cast<bf16x4*>(p + lane * 4) = a;
  *reinterpret_cast<bf16x4*>(p + 256 + lane * 4) = b;
}
// LayerNorm statistics of one 512-wide row held across the wave (two-pass, biased variance, eps 1e-5)
__device__ __forceinline__ void ln_stats(const Row8& r, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += r.a[i] + r.b[i];
  mean = wave_sum(s) * (1.f / DM);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float da = r.a[i] - mean, db = r.b[i] - mean;
    v += da * da + db * db;
  }
  rstd = 1.0f / sqrtf(wave_sum(v) * (1.f / DM) + 1e-5f);
}

// x[row] = float(double(tok_emb[id]) + (PE[t+1] + PPE[g+1]))   (PTransformer.py:659-679)
__global__ __launch_bounds__(256) void embed_pe_kernel(const int* __restrict__ ids, const float* __restrict__ tok,
                                                       const double* __restrict__ pe, const double* __restrict__ ppe,
                                                       float* __restrict__ x, unsigned short* __restrict__ xb,
                                                       int rows, int N, int G, int vocab) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int n = row % N, t = n / G, g = n - t * G;
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float* e = tok + (size_t)id * DM;
  const double* p1 = pe + (size_t)(t + 1) * DM;
  const double* p2 = ppe + (size_t)(g + 1) * DM;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int c = half * 256 + lane * 4;
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (float)((double)e[c + i] + (p1[c + i] + p2[c + i]));
    if (x) *reinterpret_cast<f32x4*>(x + (size_t)row * DM + c) = o;
    if (xb) {
      bf16x4 ob;
#pragma unroll
      for (int i = 0; i < 4; ++i) ob[i] = (__bf16)o[i];
      *reinterpret_cast<bf16x4*>(xb + (size_t)row * DM + c) = ob;
    }
  }
}
int launch_embed_pe(const int* ids, const float* tok, const double* pe, const double* ppe, float* x,
                    unsigned short* xb, int rows, int N, int G, int vocab, hipStream_t s) {
  PT_LAUNCH(embed_pe_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, ids, tok, pe, ppe, x, xb, rows, N, G, vocab);
  return 0;
}

// out = SiLU(LN(h; g, b) * (1 + scale[label]) + shift[label])     (PTransformer.py:134-139)
// style: [n_classes][2*DM] = emb_layers_2(SiLU(class_embed[c])) precomputed at load; label per clip.
__global__ __launch_bounds__(256) void ln_mod_silu_kernel(const float* __restrict__ h, float* __restrict__ out,
                                                          unsigned short* __restrict__ outb,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ style,
                                                          const int64_t* __restrict__ label, int n_label, int rows,
                                                          int rows_per_clip, int n_classes) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  Row8 r = load_row(h + (size_t)row * DM, lane);
  float mean, rstd;
  ln_stats(r, mean, rstd);
  const int clip = row / rows_per_clip;
  long long lb = label[n_label == 1 ? 0 : clip];
  lb = lb < 0 ? 0 : (lb >= n_classes ? n_classes - 1 : lb);
  const float* st = style + (size_t)lb * 2 * DM;
  const Row8 g = load_row(gamma, lane), b = load_row(beta, lane), sc = load_row(st, lane), sh = load_row(st + DM, lane);
  Row8 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float ya = ((r.a[i] - mean) * rstd * g.a[i] + b.a[i]) * (1.f + sc.a[i]) + sh.a[i];
    float yb = ((r.b[i] - mean) * rstd * g.b[i] + b.b[i]) * (1.f + sc.b[i]) + sh.b[i];
    o.a[i] = ya / (1.f + expf(-ya));
    o.b[i] = yb / (1.f + expf(-yb));
  }
  if (out) store_row(out + (size_t)row * DM, lane, o);
  if (outb) store_row_bf16(outb + (size_t)row * DM, lane, o);
}
int launch_ln_mod_silu(const float* h, float* out, unsigned short* outb, const float* gamma, const float* beta,
                       const float* style, const int64_t* label, int n_label, int rows, int rows_per_clip,
                       int n_classes, hipStream_t s) {
  PT_LAUNCH(ln_mod_silu_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, h, out, outb, gamma, beta, style, label,
                     n_label, rows, rows_per_clip, n_classes);
  return 0;
}

// x = LN(x + y; g, b)   (post-LN residual, PTransformer.py:329-331)
__global__ __launch_bounds__(256) void add_ln_kernel(float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int rows, unsigned short* __restrict__ xb) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  Row8 r = load_row(x + (size_t)row * DM, lane);
  const Row8 t = load_row(y + (size_t)row * DM, lane);
  r.a += t.a;
  r.b += t.b;
  float mean, rstd;
  ln_stats(r, mean, rstd);
  const Row8 g = load_row(gamma, lane), b = load_row(beta, lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.a[i] = (r.a[i] - mean) * rstd * g.a[i] + b.a[i];
    r.b[i] = (r.b[i] - mean) * rstd * g.b[i] + b.b[i];
  }
  store_row(x + (size_t)row * DM, lane, r);
  if (xb) store_row_bf16(xb + (size_t)row * DM, lane, r);
}
int launch_add_ln(float* x, const float* y, const float* gamma, const float* beta, int rows, unsigned short* xb,
                  hipStream_t s) {
  PT_LAUNCH(add_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, y, gamma, beta, rows, xb);
  return 0;
}

// the same with the bf16 residual stream of the bf16 mode (x lives in bf16 only): the small-batch path, where the Linear
// runs on 64 x 64 tiles and its LayerNorm cannot be an epilogue (gemm_bf16_sm.hip)
__global__ __launch_bounds__(256) void add_ln_b_kernel(unsigned short* __restrict__ xb, const float* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16x4 xa = *reinterpret_cast<const bf16x4*>(xb + (size_t)row * DM + lane * 4);
  const bf16x4 xc = *reinterpret_cast<const bf16x4*>(xb + (size_t)row * DM + 256 + lane * 4);
  Row8 r = load_row(y + (size_t)row * DM, lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.a[i] += (float)xa[i];
    r.b[i] += (float)xc[i];
  }
  float mean, rstd;
  ln_stats(r, mean, rstd);
  const Row8 g = load_row(gamma, lane), b = load_row(beta, lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.a[i] = (r.a[i] - mean) * rstd * g.a[i] + b.a[i];
    r.b[i] = (r.b[i] - mean) * rstd * g.b[i] + b.b[i];
  }
  store_row_bf16(xb + (size_t)row * DM, lane, r);
}
int launch_add_ln_b(unsigned short* xb, const float* y, const float* gamma, const float* beta, int rows, hipStream_t s) {
  PT_LAUNCH(add_ln_b_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, xb, y, gamma, beta, rows);
  return 0;
}

// refiner condition: one-hot(mask) appended as two channels (RTransformer.py:218-219):
// cond_cl[row][C0] = (mask==0), cond_cl[row][C0+1] = (mask==1), remaining pad columns zero.
__global__ void mask_onehot_kernel(const float* __restrict__ mask, float* __restrict__ cond, int rows, int ld, int c0,
                                   int zero_to) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const long long m = (long long)mask[row];
  float* p = cond + (size_t)row * ld;
  p[c0] = m == 0 ? 1.f : 0.f;
  p[c0 + 1] = m == 1 ? 1.f : 0.f;
  for (int c = c0 + 2; c < zero_to; ++c) p[c] = 0.f;
}
int launch_mask_onehot(const float* mask, float* cond, int rows, int ld, int c0, int zero_to, hipStream_t s) {
  PT_LAUNCH(mask_onehot_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, mask, cond, rows, ld, c0, zero_to);
  return 0;
}

// ---------------------------------------------------------------------------
// One MaskGIT iteration after the logits call (PTransformer.py:607-638), one workgroup per clip:
//   sampled = argmax(softmax(logits) / q)            (torch.multinomial's single-draw path)
//   conf    = log p[sampled] + temp * gumbel, +inf for tokens already known
//   re-mask every token whose confidence is below the mask_len-th smallest.
// ---------------------------------------------------------------------------
// Counter-based noise of the sampler's default mode (no noise tensors in HBM): element idx of stream `stream` under `seed`
// -> uniform in (0,1) with 23 random bits (splitmix64 finaliser of a Weyl sequence); the categorical draw uses
// q = -log u ~ Exp(1), the confidence uses g = -log(-log u) ~ Gumbel(0,1).
// k + 0.5 with k < 2^23 is exact in fp32 (24 significant bits), so u lies in [2^-24, 1 - 2^-24]: never 0 and never 1
// (with 24 bits k = 2^24 - 1 rounded to 2^24 and gave u = 1, a Gumbel draw of +inf and, at temp = 0, a NaN confidence).
__device__ __forceinline__ float mg_bits_to_uniform(unsigned k23) {
  return ((float)(k23 & 0x7FFFFFu) + 0.5f) * (1.0f / 8388608.0f);
}
__device__ __forceinline__ float mg_uniform(unsigned long long seed, unsigned long long stream, unsigned long long idx) {
  unsigned long long z = idx * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull + stream * 0x8CB92BA72F3D8DD7ull;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return mg_bits_to_uniform((unsigned)(z >> 41));
}
__device__ __forceinline__ float mg_exp1(unsigned long long seed, int step, unsigned long long e) {
  return -logf(mg_uniform(seed, 2ull * step, e));
}
__device__ __forceinline__ float mg_gumbel(unsigned long long seed, int step, unsigned long long e) {
  return -logf(-logf(mg_uniform(seed, 2ull * step + 1, e)));
}
// the noise the seeded sampler uses at `step`, written out (tests: injected mode fed with it reproduces the seeded mode)
__global__ void maskgit_noise_kernel(unsigned long long seed, int step, size_t nq, size_t ng, float* __restrict__ q,
                                     float* __restrict__ g) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nq; e += (size_t)gridDim.x * blockDim.x) {
    q[e] = mg_exp1(seed, step, e);
    if (e < ng) g[e] = mg_gumbel(seed, step, e);
  }
}
// test hook: the bits -> (u, Exp(1), Gumbel) mapping of the generator on caller-chosen bit patterns (extremes of the range)
__global__ void maskgit_bits_kernel(const unsigned* __restrict__ bits, int n, float* __restrict__ u, float* __restrict__ q,
                                    float* __restrict__ g) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = mg_bits_to_uniform(bits[i]);
  u[i] = v;
  q[i] = -logf(v);
  g[i] = -logf(-logf(v));
}
int launch_maskgit_bits(const unsigned* bits, int n, float* u, float* q, float* g, hipStream_t s) {
  PT_CHECK(bits && u && q && g && n > 0, "maskgit_bits: bad argument");
  PT_LAUNCH(maskgit_bits_kernel, dim3((n + 255) / 256), dim3(256), 0, s, bits, n, u, q, g);
  return 0;
}
int launch_maskgit_noise(unsigned long long seed, int step, int B, int N, int V, float* q, float* g, hipStream_t s) {
  PT_CHECK(B > 0 && N > 0 && V > 0 && q && g, "maskgit_noise: bad argument");
  const size_t nq = (size_t)B * N * V, ng = (size_t)B * N;
  PT_LAUNCH(maskgit_noise_kernel, dim3((unsigned)std::min<size_t>((nq + 255) / 256, 8192)), dim3(256), 0, s, seed, step, nq, ng, q, g);
  return 0;
}

__global__ __launch_bounds__(1024) void maskgit_step_kernel(const float* __restrict__ logits,
                                                           const float* __restrict__ noise_q,
                                                           const float* __restrict__ noise_g, int* __restrict__ ids,
                                                           int N, int V, int mask_id, int want_mask, float temp,
                                                           unsigned long long seed, int step,
                                                           const unsigned long long* __restrict__ seed_dev) {
  extern __shared__ float sm[];
  if (seed_dev) seed = *seed_dev;                    // seed kept in device memory: a captured graph replays with new seeds
  float* conf = sm;                                  // [N]
  int* samp = reinterpret_cast<int*>(conf + N);      // [N]
  __shared__ int unknown_cnt;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) unknown_cnt = 0;
  __syncthreads();
  int my_unknown = 0;
  const int nwaves = blockDim.x >> 6;
  for (int n = wave; n < N; n += nwaves) {
    const float* lg = logits + ((size_t)b * N + n) * V;
    const size_t qbase = ((size_t)b * N + n) * V;            // noise_q == NULL: counter-based noise (seed, step)
    float mx = -INFINITY;
    for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lg[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < V; c += 64) sum += expf(lg[c] - mx);
    sum = wave_sum(sum);
    float best = -INFINITY, bestp = 0.f;
    int besti = 0x7fffffff;
    for (int c = lane; c < V; c += 64) {
      const float p = expf(lg[c] - mx) / sum;
      const float r = p / (noise_q ? noise_q[qbase + c] : mg_exp1(seed, step, qbase + c));
      if (r > best) { best = r; besti = c; bestp = p; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ob = __shfl_xor(best, o), op = __shfl_xor(bestp, o);
      const int oi = __shfl_xor(besti, o);
      if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; bestp = op; }
    }
    if (lane == 0) {
      const int cur = ids[(size_t)b * N + n];
      const bool unk = cur == mask_id;
      samp[n] = unk ? besti : cur;
      const float gn = noise_g ? noise_g[(size_t)b * N + n] : mg_gumbel(seed, step, (size_t)b * N + n);
      conf[n] = unk ? (temp != 0.f ? logf(bestp) + temp * gn : logf(bestp)) : INFINITY;
      my_unknown += unk;
    }
  }
  if (lane == 0 && my_unknown) atomicAdd(&unknown_cnt, my_unknown);
  __syncthreads();
  int k = unknown_cnt - 1;
  k = k < want_mask ? k : want_mask;
  k = k < 0 ? 0 : k;
  for (int n = tid; n < N; n += blockDim.x) {
    const float c = conf[n];
    int le = 0;
    for (int m = 0; m < N; ++m) le += conf[m] <= c;
    ids[(size_t)b * N + n] = (le <= k) ? mask_id : samp[n];
  }
}
int launch_maskgit_step(const float* logits, const float* noise_q, const float* noise_g, int* ids, int B, int N, int V,
                        int mask_id, int want_mask, float temp, hipStream_t s, unsigned long long seed, int step,
                        const unsigned long long* seed_dev) {
  PT_CHECK(N > 0 && N <= 8192, "maskgit_step: N=%d out of range", N);
  PT_CHECK((noise_q == nullptr) == (noise_g == nullptr), "maskgit_step: both noise tensors or neither");
  // few clips: 16 waves per clip (one workgroup per clip is all the parallelism there is); many clips: 4 waves, 4+ clips per CU
  PT_LAUNCH(maskgit_step_kernel, dim3(B), dim3(B <= 256 ? 1024 : 256), (size_t)N * 8, s, logits, noise_q, noise_g, ids, N, V,
                     mask_id, want_mask, temp, seed, step, seed_dev);
  return 0;
}

__global__ void fill_i32_kernel(int* p, int v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int launch_fill_i32(int* p, int v, size_t n, hipStream_t s) {
  PT_LAUNCH(fill_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
  return 0;
}

// ---------------------------------------------------------------------------
// F.interpolate(mode='linear', align_corners=False) along time + mask blend, NCL in / NCL out.
// ---------------------------------------------------------------------------
// One frame: v = interpolated (or copied, or absent) prediction, blended with gt under the mask.
__device__ __forceinline__ float interp_blend_one(float v, bool has_gt, float g, float m, int premask) {
  if (has_gt) {
    float a = g * m;
    if (premask) a = a * m;
    v = a + v * (1.f - m);
  }
  return v;
}
// General case (time is resampled, or an odd length / alignment): one thread per output frame, 32-bit index arithmetic (rows * T_out < 2^32 checked by
// the launcher).
__global__ __launch_bounds__(256) void interp_blend_kernel(const float* __restrict__ pred, int T_in, int T_out,
                                                           const float* __restrict__ gt,
                                                           const float* __restrict__ mask, int premask,
                                                           float* __restrict__ out, int C, unsigned total) {
  const float scale = (float)T_in / (float)T_out;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    const unsigned bc = e / (unsigned)T_out, t = e - bc * (unsigned)T_out;
    float v = 0.f;
    if (pred != nullptr) {
      const float* prow = pred + (size_t)bc * T_in;
      if (T_in == T_out) {
        v = prow[t];
      } else {
        float src = scale * ((float)t + 0.5f) - 0.5f;
        src = src < 0.f ? 0.f : src;
        const int i0 = (int)src;
        const int i1 = i0 + (i0 < T_in - 1 ? 1 : 0);
        const float l1 = src - (float)i0, l0 = 1.f - l1;
        v = l0 * prow[i0] + l1 * prow[i1];
      }
    }
    const bool has_gt = gt != nullptr;
    out[e] = interp_blend_one(v, has_gt, has_gt ? gt[e] : 0.f, has_gt ? mask[(bc / (unsigned)C) * (unsigned)T_out + t] : 0.f, premask);
  }
}
// T_in == T_out (or no prediction at all) and T_out % 4 == 0: pure streaming, four frames (16 bytes) per thread and
// tensor.  This is what the pipeline calls 5 times out of 6 (300-frame clips): 350 MB per call at configs[1]'s half batch.
__global__ __launch_bounds__(256) void blend4_kernel(const float* __restrict__ pred, int T, const float* __restrict__ gt,
                                                     const float* __restrict__ mask, int premask, float* __restrict__ out,
                                                     int C, unsigned total4) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const unsigned T4 = (unsigned)T >> 2;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total4; e += gridDim.x * 256u) {
    const unsigned bc = e / T4, t4 = e - bc * T4;
    const size_t o = (size_t)e * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (pred) v = *reinterpret_cast<const f32x4*>(pred + o);
    if (gt) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gt + o);
      const f32x4 m = *reinterpret_cast<const f32x4*>(mask + ((size_t)(bc / (unsigned)C) * T4 + t4) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = interp_blend_one(v[i], true, g[i], m[i], premask);
    }
    *reinterpret_cast<f32x4*>(out + o) = v;
  }
}
int launch_interp_blend(const float* pred, int B, int C, int T_in, int T_out, const float* gt, const float* mask,
                        int premask, float* out, hipStream_t s) {
  PT_CHECK(B > 0 && C > 0 && T_in > 0 && T_out > 0, "interp_blend: empty tensor");
  PT_CHECK((gt == nullptr) == (mask == nullptr), "interp_blend: gt and mask go together");
  const size_t total = (size_t)B * C * T_out;
  // the kernels' 32-bit grid-stride counters advance by up to 8192 * 256 per step: the limit leaves that much headroom
  // below 2^32 so that the last increment cannot wrap
  PT_CHECK(total <= (1ull << 32) - 8192ull * 256ull, "interp_blend: %zu output elements (limit 2^32 - 2^21 per call)", total);
  const bool aligned = ((((uintptr_t)pred | (uintptr_t)gt | (uintptr_t)mask | (uintptr_t)out) & 15) == 0);
  if ((pred == nullptr || T_in == T_out) && T_out % 4 == 0 && aligned) {
    const unsigned total4 = (unsigned)(total / 4);
    const unsigned blocks = std::min<unsigned>((total4 + 255) / 256, 8192);
    PT_LAUNCH(blend4_kernel, dim3(blocks), dim3(256), 0, s, pred, T_out, gt, mask, premask, out, C, total4);
    return 0;
  }
  const unsigned blocks = std::min<unsigned>((unsigned)((total + 255) / 256), 8192);
  PT_LAUNCH(interp_blend_kernel, dim3(blocks), dim3(256), 0, s, pred, T_in, T_out, gt, mask, premask, out, C, (unsigned)total);
  return 0;
}

}  // namespace pt
