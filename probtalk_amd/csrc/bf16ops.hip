// bf16-mode helpers of the transformer path: fp32->bf16 row conversion and the attention core on
// v_mfma_f32_32x32x16_bf16 (see the comment above attention_bf16_kernel).
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

// ---------------------------------------------------------------------------
// Attention core softmax(Q K^T / 8) V on v_mfma_f32_32x32x16_bf16, one workgroup per (clip, head).
//   * K and V of the head (Nk x 64 each) are staged ONCE into LDS by LDS-DMA (global_load_lds_dwordx4) and shared
//     by the workgroup's waves; each wave owns 32-query tiles.
//   * S^T = K Q^T ("swapped"): one query per lane column, so softmax runs in registers and the probabilities,
//     packed pairwise to bf16, ARE the B operand of O^T += V^T P^T (cdna guide §3, accumulator tile as operand:
//     element jj of lane half h in k-step s is key 16s + 8(jj>>2) + 4h + (jj&3)).
//   * the matching A operand (V^T: a lane = one feature d, 4+4 consecutive keys) comes from the row-major V
//     image through ds_read_b64_tr_b16, the hardware transposing read: no V^T copy in HBM any more.
//   * LDS images are lane-linear per DMA instruction (8 rows x 128 B); bank swizzles live in the per-lane source
//     address: K chunk ^= (row>>1)&7 (conflict-free ds_read_b128), V chunk ^= ((row>>1)&1)<<2 (conflict-free
//     transposing reads).
//   * CT key tiles (32 keys each) are held in registers per softmax step; longer key ranges continue with the
//     usual running max / sum rescale.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

template <int CT, int MAXT, int WPE>
__global__ __launch_bounds__(MAXT, WPE) void attention_bf16_kernel(const unsigned short* __restrict__ Q, int ldq,
                                                             const unsigned short* __restrict__ K, int ldk,
                                                             const unsigned short* __restrict__ V, int ldv,
                                                             unsigned short* __restrict__ O, int ldo, int Nq, int Nk,
                                                             int Nkp, int H, int qtiles, float scale) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char att_smem[];
  unsigned char* Ks = att_smem;
  unsigned char* Vs = att_smem + (size_t)Nkp * 128;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int head = blockIdx.x % H, b = blockIdx.x / H;
  {
    const unsigned short* kb = K + (size_t)b * Nk * ldk + head * 64;
    const unsigned short* vb = V + (size_t)b * Nk * ldv + head * 64;
    for (int R0 = wave * 8; R0 < Nkp; R0 += nw * 8) {
      const int row = R0 + (lane >> 3);
      const int srow = row < Nk ? row : Nk - 1;
      const int gk = (lane & 7) ^ ((row >> 1) & 7);
      const int gv = (lane & 7) ^ (((row >> 1) & 1) << 2);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(kb + (size_t)srow * ldk + gk * 8), (lds_ptr_t)(Ks + R0 * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(vb + (size_t)srow * ldv + gv * 8), (lds_ptr_t)(Vs + R0 * 128), 16, 0, 0);
    }
  }
  const int j = lane & 31, h2 = lane >> 5;
  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = j * 128 + (((2 * s + h2) ^ ((j >> 1) & 7)) << 4);
  // transposing read: lane 4q+p of a 16-lane group addresses key row q, features 4p..4p+3 of the group's 16
  const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
  const int voff0 = (4 * h2 + tq) * 128 + (((tg * 2 + (tp >> 1)) ^ ((tq >> 1) << 2)) << 4) + (tp & 1) * 8;
  const int voff1 = voff0 ^ 64;
  // wave-private 32-query tile image: first the Q rows (DMA, 128-byte rows, K's swizzle), later the output tile
  // (144-byte rows) on its way to row-contiguous stores
  constexpr int QO = 32 * 144;
  unsigned char* Qs = att_smem + (size_t)Nkp * 256 + wave * QO;
  auto stage_q = [&](int qt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 8 + (lane >> 3);
      int qrow = qt * 32 + row;
      qrow = qrow < Nq ? qrow : Nq - 1;
      const int g = (lane & 7) ^ ((row >> 1) & 7);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(Q + (size_t)(b * Nq + qrow) * ldq + head * 64 + g * 8),
                                       (lds_ptr_t)(Qs + i * 1024), 16, 0, 0);
    }
  };
  stage_q(wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int qt = wave; qt < qtiles; qt += nw) {
    if (qt != wave) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the previous tile's output reads have left the image
      stage_q(qt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qs + koff[s]);
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    for (int c0 = 0; c0 < Nkp; c0 += 32 * CT) {
      f32x16 st[CT];
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (c0 + 32 * t < Nkp) {
#pragma unroll
          for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (c0 + 32 * t) * 128 + koff[s]);
            st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[t], 0, 0, 0);
          }
          // raw logits here; scale * log2 e is folded into the exponent's fma below (scale > 0, so the maximum commutes);
          // only a partial last tile needs the key mask
          if (c0 + 32 * t + 32 <= Nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = c0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h2;
              st[t][r] = key < Nk ? st[t][r] : -INFINITY;
              mx = fmaxf(mx, st[t][r]);
            }
          }
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32)) * scale;
      const float m_new = fmaxf(m_run, mx);
      const float alpha = exp2f(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (c0 + 32 * t < Nkp) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            st[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][r], scale, -m_new));
            psum += st[t][r];
          }
        }
      }
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;
      m_run = m_new;
      if (c0 > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (c0 + 32 * t < Nkp) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pb;
#pragma unroll
            for (int e = 0; e < 8; ++e) pb[e] = (__bf16)st[t][8 * s2 + e];
            const unsigned char* vrow = Vs + (c0 + 32 * t + 16 * s2) * 128;
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vrow + voff0));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vrow + 8 * 128 + voff0));
            const s16x4 c0v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vrow + voff1));
            const s16x4 c1v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vrow + 8 * 128 + voff1));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 va = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const s16x8 vc = {c0v[0], c0v[1], c0v[2], c0v[3], c1v[0], c1v[1], c1v[2], c1v[3]};
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), pb, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vc), pb, o1, 0, 0, 0);
          }
        }
      }
    }
    // O^T (lane = query, registers = features) -> [query][64 features] image -> 128-byte row stores
    const float inv = 1.f / l_run;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 a, c;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = (__bf16)(o0[4 * g + e] * inv); c[e] = (__bf16)(o1[4 * g + e] * inv); }
      *reinterpret_cast<bf16x4*>(Qs + j * 144 + (8 * g + 4 * h2) * 2) = a;
      *reinterpret_cast<bf16x4*>(Qs + j * 144 + (32 + 8 * g + 4 * h2) * 2) = c;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 8 + (lane >> 3);
      const u32x4 v = *reinterpret_cast<const u32x4*>(Qs + row * 144 + (lane & 7) * 16);
      const int qrow = qt * 32 + row;
      if (qrow < Nq) *reinterpret_cast<u32x4*>(O + (size_t)(b * Nq + qrow) * ldo + head * 64 + (lane & 7) * 8) = v;
    }
  }
}

// The same core for key ranges beyond the LDS-resident limit (whole-utterance refiner calls, up to the 1,800 rows of the
// positional table): K and V pass through LDS in chunks of AS_KC keys, every wave keeps ONE 32-query tile's running
// (max, sum, O^T) in registers across the chunks, and blockIdx.y walks blocks of up to 10 query tiles.
constexpr int AS_KC = 256;
template <int CT>
__global__ __launch_bounds__(640, 3) void attention_bf16_stream_kernel(const unsigned short* __restrict__ Q, int ldq,
                                                                       const unsigned short* __restrict__ K, int ldk,
                                                                       const unsigned short* __restrict__ V, int ldv,
                                                                       unsigned short* __restrict__ O, int ldo, int Nq,
                                                                       int Nk, int H, int qtiles, float scale) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char att_smem[];
  unsigned char* Ks = att_smem;
  unsigned char* Vs = att_smem + AS_KC * 128;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int head = blockIdx.x % H, b = blockIdx.x / H;
  const int qt_raw = blockIdx.y * nw + wave;
  const bool active = qt_raw < qtiles;                 // idle waves still stage K / V and meet the barriers
  const int qt = active ? qt_raw : qtiles - 1;
  const unsigned short* kb = K + (size_t)b * Nk * ldk + head * 64;
  const unsigned short* vb = V + (size_t)b * Nk * ldv + head * 64;
  const int j = lane & 31, h2 = lane >> 5;
  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = j * 128 + (((2 * s + h2) ^ ((j >> 1) & 7)) << 4);
  const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
  const int voff0 = (4 * h2 + tq) * 128 + (((tg * 2 + (tp >> 1)) ^ ((tq >> 1) << 2)) << 4) + (tp & 1) * 8;
  const int voff1 = voff0 ^ 64;
  constexpr int QO = 32 * 144;
  unsigned char* Qs = att_smem + AS_KC * 256 + wave * QO;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 8 + (lane >> 3);
    int qrow = qt * 32 + row;
    qrow = qrow < Nq ? qrow : Nq - 1;
    const int g = (lane & 7) ^ ((row >> 1) & 7);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(Q + (size_t)(b * Nq + qrow) * ldq + head * 64 + g * 8),
                                     (lds_ptr_t)(Qs + i * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // wave-private image: no barrier needed
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qs + koff[s]);
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < Nk; k0 += AS_KC) {
    const int nkc = (Nk - k0 < AS_KC ? ((Nk - k0 + 31) & ~31) : AS_KC);        // keys of this chunk, padded to tiles
    __syncthreads();                                     // every wave is done with the previous chunk
    for (int R0 = wave * 8; R0 < nkc; R0 += nw * 8) {
      const int lrow = R0 + (lane >> 3);
      const int srow = k0 + lrow < Nk ? k0 + lrow : Nk - 1;
      const int gk = (lane & 7) ^ ((lrow >> 1) & 7);
      const int gv = (lane & 7) ^ (((lrow >> 1) & 1) << 2);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(kb + (size_t)srow * ldk + gk * 8), (lds_ptr_t)(Ks + R0 * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(vb + (size_t)srow * ldv + gv * 8), (lds_ptr_t)(Vs + R0 * 128), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c0 = 0; c0 < nkc; c0 += 32 * CT) {
      f32x16 st[CT];
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (c0 + 32 * t < nkc) {
#pragma unroll
          for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (c0 + 32 * t) * 128 + koff[s]);
            st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[t], 0, 0, 0);
          }
          if (k0 + c0 + 32 * t + 32 <= Nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = k0 + c0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h2;
              st[t][r] = key < Nk ? st[t][r] : -INFINITY;
              mx = fmaxf(mx, st[t][r]);
            }
          }
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32)) * scale;
      const float m_new = fmaxf(m_run, mx);
      const float alpha = exp2f(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (c0 + 32 * t < nkc) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            st[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][r], scale, -m_new));
            psum += st[t][r];
          }
        }
      }
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;
      m_run = m_new;
      if (k0 + c0 > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (c0 + 32 * t < nkc) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pb;
#pragma unroll
            for (int e = 0; e < 8; ++e) pb[e] = (__bf16)st[t][8 * s2 + e];
            const unsigned char* vrow = Vs + (c0 + 32 * t + 16 * s2) * 128;
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vrow + voff0));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vrow + 8 * 128 + voff0));
            const s16x4 c0v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vrow + voff1));
            const s16x4 c1v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vrow + 8 * 128 + voff1));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 va = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const s16x8 vc = {c0v[0], c0v[1], c0v[2], c0v[3], c1v[0], c1v[1], c1v[2], c1v[3]};
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), pb, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vc), pb, o1, 0, 0, 0);
          }
        }
      }
    }
  }
  const float inv = 1.f / l_run;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4 a, c;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = (__bf16)(o0[4 * g + e] * inv); c[e] = (__bf16)(o1[4 * g + e] * inv); }
    *reinterpret_cast<bf16x4*>(Qs + j * 144 + (8 * g + 4 * h2) * 2) = a;
    *reinterpret_cast<bf16x4*>(Qs + j * 144 + (32 + 8 * g + 4 * h2) * 2) = c;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 8 + (lane >> 3);
    const u32x4 v = *reinterpret_cast<const u32x4*>(Qs + row * 144 + (lane & 7) * 16);
    const int qrow = qt * 32 + row;
    if (active && qrow < Nq) *reinterpret_cast<u32x4*>(O + (size_t)(b * Nq + qrow) * ldo + head * 64 + (lane & 7) * 8) = v;
  }
}

int launch_attention_bf16(const unsigned short* Q, int ldq, const unsigned short* K, int ldk, const unsigned short* V,
                          int ldv, unsigned short* O, int ldo, int B, int Nq, int Nk, int H, hipStream_t s) {
  PT_CHECK(B > 0 && Nq > 0 && Nk > 0 && H > 0, "attention_bf16: empty problem");
  PT_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "attention_bf16: leading dims must be multiples of 8");
  PT_CHECK((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) == 0,
           "attention_bf16: operands must be 16-byte aligned");
  const int Nkp = round_up(Nk, 32);
  const int qtiles = (Nq + 31) / 32;
  const int nw = qtiles <= 10 ? qtiles : (qtiles + 1) / 2 <= 10 ? (qtiles + 1) / 2 : 10;   // waves = query tiles when they fit
  constexpr int CT = 2;
  const size_t smem = (size_t)Nkp * 256 + (size_t)nw * 32 * 144;
  if (smem > 160 * 1024) {
    // beyond the LDS-resident limit (544 keys with 10 waves): K / V stream through LDS in chunks
    PT_ENSURE_LDS((attention_bf16_stream_kernel<CT>), 160 * 1024);
    const int snw = qtiles < 10 ? qtiles : 10;
    const size_t ssmem = (size_t)AS_KC * 256 + (size_t)snw * 32 * 144;
    PT_LAUNCH((attention_bf16_stream_kernel<CT>), dim3(B * H, (qtiles + snw - 1) / snw), dim3(64 * snw), ssmem, s, Q, ldq, K, ldk,
              V, ldv, O, ldo, Nq, Nk, H, qtiles, 0.125f * 1.4426950408889634f);
    return 0;
  }
  PT_ENSURE_LDS((attention_bf16_kernel<CT, 320, 4>), 160 * 1024);
  PT_ENSURE_LDS((attention_bf16_kernel<CT, 640, 3>), 160 * 1024);
  // up to 5 query tiles: 5-wave workgroups, 128 registers (two workgroups per CU at 148 keys); longer sequences:
  // up to 10 waves sharing one K/V image, 168 registers
  if (nw <= 5)
    PT_LAUNCH((attention_bf16_kernel<CT, 320, 4>), dim3(B * H), dim3(64 * nw), smem, s, Q, ldq, K, ldk, V, ldv, O, ldo, Nq,
              Nk, Nkp, H, qtiles, 0.125f * 1.4426950408889634f);
  else
    PT_LAUNCH((attention_bf16_kernel<CT, 640, 3>), dim3(B * H), dim3(64 * nw), smem, s, Q, ldq, K, ldk, V, ldv, O, ldo, Nq,
              Nk, Nkp, H, qtiles, 0.125f * 1.4426950408889634f);
  return 0;
}

}  // namespace pt
