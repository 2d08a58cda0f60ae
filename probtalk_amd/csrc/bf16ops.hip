// bf16-mode helpers of the transformer path: fp32->bf16 row conversion, the per-head V transpose and
// the attention core on v_mfma_f32_32x32x16_bf16.
//
// Attention uses the same "swapped" formulation as attention.hip: S^T = K Q^T puts one query per lane
// column, so softmax is in-register and the fp32 probabilities, packed pairwise to bf16, ARE the B
// operand of O^T += V^T P^T (cdna guide §3 "an accumulator tile as the next MFMA's operand": element
// jj of lane half h in k-step s is key 16s + 8(jj>>2) + 4h + (jj&3)).  The matching A operand needs,
// per lane (= one output feature d), 4+4 consecutive keys of V^T; V is therefore transposed per head
// once (vt_kernel) so those are two 8-byte loads.  No LDS, no barriers: waves are independent.
#include "transformer.h"

namespace pt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cvt_f32_bf16_kernel(const float* __restrict__ in, int ldi,
                                                           unsigned short* __restrict__ out, int ldo, int rows,
                                                           int cols, int zero_to) {
  const int per_row = zero_to / 4;
  const size_t total = (size_t)rows * per_row;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int row = (int)(e / per_row), c = (int)(e - (size_t)row * per_row) * 4;
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (__bf16)((c + i < cols) ? in[(size_t)row * ldi + c + i] : 0.f);
    *reinterpret_cast<bf16x4*>(out + (size_t)row * ldo + c) = o;
  }
}
int launch_cvt_f32_bf16(const float* in, int ldi, unsigned short* out, int ldo, int rows, int cols, int zero_to,
                        hipStream_t s) {
  PT_CHECK(rows > 0 && cols > 0 && zero_to % 4 == 0 && zero_to >= cols && zero_to <= ldo && ldo % 4 == 0,
           "cvt_f32_bf16: bad geometry rows=%d cols=%d zero_to=%d ldo=%d", rows, cols, zero_to, ldo);
  const size_t total = (size_t)rows * (zero_to / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  PT_LAUNCH(cvt_f32_bf16_kernel, dim3(blocks), dim3(256), 0, s, in, ldi, out, ldo, rows, cols, zero_to);
  return 0;
}

// V (rows b*Nk + key, 64 features of head h at V + h*64) -> VT[b][h][d][Nkp], keys >= Nk zero filled
__global__ __launch_bounds__(256) void vt_kernel(const unsigned short* __restrict__ V, int ldv,
                                                 unsigned short* __restrict__ VT, int Nk, int Nkp, int H) {
  __shared__ unsigned short tile[32][72];
  const int k0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  {
    const int key = tid >> 3, d8 = (tid & 7) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (k0 + key < Nk) v = *reinterpret_cast<const u32x4*>(V + (size_t)(b * Nk + k0 + key) * ldv + h * 64 + d8);
    *reinterpret_cast<u32x4*>(&tile[key][d8]) = v;
  }
  __syncthreads();
  {
    const int d = tid >> 2, k8 = (tid & 3) * 8;
    unsigned short o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = tile[k8 + i][d];
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (unsigned)o[2 * i] | ((unsigned)o[2 * i + 1] << 16);
    *reinterpret_cast<u32x4*>(VT + ((size_t)(b * H + h) * 64 + d) * Nkp + k0 + k8) = v;
  }
}
int launch_vt(const unsigned short* V, int ldv, unsigned short* VT, int B, int Nk, int H, hipStream_t s) {
  PT_CHECK(B > 0 && Nk > 0 && ldv % 8 == 0 && ((uintptr_t)V & 15) == 0, "vt: bad geometry");
  const int Nkp = round_up(Nk, 32);
  PT_LAUNCH(vt_kernel, dim3(Nkp / 32, H, B), dim3(256), 0, s, V, ldv, VT, Nk, Nkp, H);
  return 0;
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_bf16_kernel(const unsigned short* __restrict__ Q, int ldq,
                                                             const unsigned short* __restrict__ K, int ldk,
                                                             const unsigned short* __restrict__ VT,
                                                             unsigned short* __restrict__ O, int ldo, int Nq, int Nk,
                                                             int Nkp, int H, int qtiles, int total_waves,
                                                             float scale) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= total_waves) return;
  const int lane = threadIdx.x & 63, j = lane & 31, h2 = lane >> 5;
  const int qt = wid % qtiles, bh = wid / qtiles, head = bh % H, b = bh / H;
  const int qrow = qt * 32 + j;
  const bool qvalid = qrow < Nq;
  const unsigned short* qp = Q + (size_t)(b * Nq + (qvalid ? qrow : Nq - 1)) * ldq + head * 64 + 8 * h2;
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const unsigned short* kbase = K + (size_t)b * Nk * ldk + head * 64 + 8 * h2;
  const unsigned short* vt0 = VT + ((size_t)(b * H + head) * 64 + j) * Nkp + 4 * h2;   // d = j
  const unsigned short* vt1 = vt0 + (size_t)32 * Nkp;                                  // d = 32 + j

  for (int k0 = 0; k0 < Nk; k0 += 32) {
    const int krow = k0 + j < Nk ? k0 + j : Nk - 1;
    const unsigned short* kp = kbase + (size_t)krow * ldk;
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + 16 * s);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h2;
      st[r] = key < Nk ? st[r] * scale : -INFINITY;
      mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = __expf(st[r] - m_new);
      psum += st[r];
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 pb;
#pragma unroll
      for (int e = 0; e < 8; ++e) pb[e] = (__bf16)st[8 * s2 + e];
      // keys k0 + 16 s2 + 4 h2 + {0..3} and + 8 + {0..3}: two 8-byte pieces of this lane's V^T row
      const u32x2 a0 = *reinterpret_cast<const u32x2*>(vt0 + k0 + 16 * s2);
      const u32x2 a1 = *reinterpret_cast<const u32x2*>(vt0 + k0 + 16 * s2 + 8);
      const u32x2 c0 = *reinterpret_cast<const u32x2*>(vt1 + k0 + 16 * s2);
      const u32x2 c1 = *reinterpret_cast<const u32x2*>(vt1 + k0 + 16 * s2 + 8);
      const u32x4 va = {a0[0], a0[1], a1[0], a1[1]}, vc = {c0[0], c0[1], c1[0], c1[1]};
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), pb, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vc), pb, o1, 0, 0, 0);
    }
  }
  if (!qvalid) return;
  const float inv = 1.f / l_run;
  unsigned short* op = O + (size_t)(b * Nq + qrow) * ldo + head * 64 + 4 * h2;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4 a, c;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = (__bf16)(o0[4 * g + e] * inv); c[e] = (__bf16)(o1[4 * g + e] * inv); }
    *reinterpret_cast<bf16x4*>(op + 8 * g) = a;
    *reinterpret_cast<bf16x4*>(op + 32 + 8 * g) = c;
  }
}

int launch_attention_bf16(const unsigned short* Q, int ldq, const unsigned short* K, int ldk,
                          const unsigned short* VT, unsigned short* O, int ldo, int B, int Nq, int Nk, int H,
                          hipStream_t s) {
  PT_CHECK(B > 0 && Nq > 0 && Nk > 0 && H > 0, "attention_bf16: empty problem");
  PT_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0, "attention_bf16: leading dims must be multiples of 8");
  PT_CHECK((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)VT) & 15) == 0 && ((uintptr_t)O & 7) == 0,
           "attention_bf16: operands must be 16-byte aligned");
  const int qtiles = (Nq + 31) / 32, total = B * H * qtiles;
  PT_LAUNCH(attention_bf16_kernel, dim3((total + 3) / 4), dim3(256), 0, s, Q, ldq, K, ldk, VT, O, ldo, Nq, Nk,
            round_up(Nk, 32), H, qtiles, total, 0.125f);
  return 0;
}

}  // namespace pt
