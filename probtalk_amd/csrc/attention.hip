// Multi-head attention core (softmax(Q K^T / sqrt(64)) V, 8 heads x 64) for the short sequences of
// the predictor (148 tokens x 148 / x 37 memory) and refiner (<= 300 x 300) on the fp32 matrix cores.
//
// One wave owns 32 queries of one (clip, head) and walks the keys in tiles of 32 with an online
// softmax.  The score tile is computed TRANSPOSED, S^T = K Q^T, so that after the MFMA each lane
// holds 16 keys of ONE query column: the row max / sum are 16 in-register ops plus one cross-half
// exchange, and the probabilities are already the B operand (k = key, col = query) of
// O^T += V^T P^T with no LDS round trip.  fp32 MFMA (64 cycles per 32x32x2) is slow enough that the
// K/V fragments are read straight from L2 (one dword(x4) per lane per MFMA), so waves share nothing
// and need no barrier.
#include "transformer.h"

namespace pt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64;

__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ Q, int ldq,
                                                            const float* __restrict__ K, int ldk,
                                                            const float* __restrict__ V, int ldv,
                                                            float* __restrict__ O, int ldo, int Nq, int Nk, int H,
                                                            int qtiles, int total_waves, float scale) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= total_waves) return;
  const int lane = threadIdx.x & 63, j = lane & 31, h2 = lane >> 5;
  const int qt = wid % qtiles, bh = wid / qtiles, head = bh % H, b = bh / H;
  const int q0 = qt * 32;
  const int qrow = q0 + j;
  const bool qvalid = qrow < Nq;
  const float* qp = Q + (size_t)(b * Nq + (qvalid ? qrow : Nq - 1)) * ldq + head * HD + 4 * h2;
  f32x4 qf[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) qf[u] = *reinterpret_cast<const f32x4*>(qp + 8 * u);

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  const float* kbase = K + (size_t)b * Nk * ldk + head * HD + 4 * h2;
  const float* vbase = V + (size_t)b * Nk * ldv + head * HD + j;

  for (int k0 = 0; k0 < Nk; k0 += 32) {
    // ---- S^T tile: rows = keys (A operand from K), cols = queries (B operand from Q)
    const int krow = k0 + j < Nk ? k0 + j : Nk - 1;
    const float* kp = kbase + (size_t)krow * ldk;
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 8 * u);
#pragma unroll
      for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[u][e], st, 0, 0, 0);
    }
    // ---- online softmax over this lane's 16 keys (+ the other half-wave's 16)
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h2;
      st[r] = key < Nk ? st[r] * scale : -INFINITY;
      mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);  // first tile: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = expf(st[r] - m_new);
      psum += st[r];
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    // ---- O^T += V^T P^T : A[i = d][k = key] from V, B[k = key][j = query] = st[r]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h2;
      key = key < Nk ? key : Nk - 1;
      const float* vp = vbase + (size_t)key * ldv;
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], st[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], st[r], o1, 0, 0, 0);
    }
  }
  if (!qvalid) return;
  const float inv = 1.f / l_run;
  float* op = O + (size_t)(b * Nq + qrow) * ldo + head * HD + 4 * h2;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 a, c;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = o0[4 * g + e] * inv; c[e] = o1[4 * g + e] * inv; }
    *reinterpret_cast<f32x4*>(op + 8 * g) = a;        // d = 8g + 4h2 + e
    *reinterpret_cast<f32x4*>(op + 32 + 8 * g) = c;   // d = 32 + 8g + 4h2 + e
  }
}

int launch_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                     int B, int Nq, int Nk, int H, hipStream_t s) {
  PT_CHECK(B > 0 && Nq > 0 && Nk > 0 && H > 0, "attention: empty problem B=%d Nq=%d Nk=%d", B, Nq, Nk);
  PT_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && ldo % 4 == 0, "attention: leading dims must be multiples of 4");
  PT_CHECK((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)O) & 15) == 0, "attention: Q/K/O must be 16-byte aligned");
  const int qtiles = (Nq + 31) / 32;
  const int total = B * H * qtiles;
  PT_LAUNCH(attention_f32_kernel, dim3((total + 3) / 4), dim3(256), 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, Nq,
                     Nk, H, qtiles, total, 0.125f);
  return 0;
}

}  // namespace pt
