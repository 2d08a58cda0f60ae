// Fused tail of a decoder sub-layer, bf16 mode (DeepSupervisionLayer, PTransformer.py:134-140,329-331 and the Linear in
// front of it), one launch instead of two:
//
//   t     = A . W1^T + b1                       last Linear of the sub-layer (attention out-proj, K1 = 512; FFN-2, K1 = 2048)
//   h     = SiLU(LN(t) (1 + scale_c) + shift_c)  stylization pre-op, c = label of the row's clip
//   x_new = LN(x + h . W2^T + b2; gamma, beta)   style Linear + post-LN residual, x in place (bf16 residual stream)
//
// Rows never leave the registers between the products.  A workgroup is 4 waves, one per SIMD (up to 512 registers per
// lane); wave w owns rows [32 w, 32 w + 32) of a 128-row tile and ALL 512 columns: 2 x 32 accumulators of
// v_mfma_f32_16x16x32_bf16 with the operands swapped (W fragment first), so a lane holds, for row m = 16 mb + (lane & 15),
// the features n = 16 nb + 4 (lane >> 4) + e.  That is already the B-operand layout of the next product up to a
// permutation of k inside each 32-block (slot 8 q + j <-> feature 4 q + j for j < 4, 16 + 4 q + (j - 4) otherwise), and
// since k is summed over, the permutation is applied to the K axis of W2 once at load time (pt_permute_k32) instead
// of to the data: no LDS round trip, no lane movement.  LayerNorm statistics are in-lane sums + two shuffles.
//
// What streams through LDS is W (one K-step of 32 = 512 rows x 64 B = 32 KB, read by all four waves) and, in the first
// product, A (128 rows x 64 B), both by LDS-DMA into a 3-slot ring: a step's image is issued two steps ahead, retired by
// a counted vmcnt at the end of the step before it is read, and read after that step's barrier.  The stream runs across
// the products and across tiles (persistent workgroups).  64-byte image rows, chunk swizzle c ^ g[(row >> 2) & 3] on the
// source address and on the fragment reads (as gemm_bf16_row.hip).
#include <algorithm>
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

constexpr int SB_W = 32768;                        // W image of one K-step: 512 rows x 64 B
constexpr int SB_A = 8192;                         // A image: 128 rows x 64 B
constexpr int SB_STAGE = SB_W + SB_A;
constexpr int SB_TAB1 = 3 * SB_STAGE;              // stylization tables [4 classes][gamma' 512 | beta' 512] floats
constexpr int SB_TAB2 = SB_TAB1 + 4 * 4096;        // post-LN gamma | beta
constexpr int SB_BIAS = SB_TAB2 + 4096;             // b1 | b2 floats
constexpr int SB_CLS = SB_BIAS + 2 * 2048;          // style class of every clip, one byte each (SB_MAXLAB clips)
constexpr int SB_MAXLAB = 2048;
constexpr int SB_SMEM = SB_CLS + SB_MAXLAB;

__device__ __forceinline__ unsigned sb_pack(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// lane id without a live register behind it: computed where it is needed (two VALU operations), so that nothing
// lane-derived stays live across the tile loop for the register allocator to spill
__device__ __forceinline__ int sb_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

__global__ __launch_bounds__(256) void sublayer_kernel(const SubArgs p, const int ntiles) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int G = gridDim.x;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt1 = p.K1 >> 5;
  const int S = nt1 + 16;                           // K-steps per tile over all products

  auto gsw = [](int k) { return (0x78 >> (2 * k)) & 3; };   // g = {0, 2, 3, 1}
  // ---- DMA streams.  One wave-instruction = 16 image rows x 64 B; lane -> row (lane >> 2), LDS chunk (lane & 3).
  // Wave w fills W image rows [128 w, 128 w + 128) with 8 instructions and its own 32 A rows with 2.  The per-lane source
  // offsets are recomputed from the lane id in every step (a handful of VALU operations): kept live across the tile loop
  // they are the first thing the register allocator spills, and a scratch reload in front of every DMA costs a vmcnt(0).
  int c_tile = blockIdx.x, c_v = 0, c_slot = 0;     // producer cursor: tile, K-step inside the tile, ring slot
  // issue context of the current step (wave-uniform)
  // (buffer_load ... lds, not global_load_lds: hipcc books the FLAT-encoded form on lgkmcnt as well and then waits
  // lgkmcnt(0) in front of every MFMA group, i.e. for the fragment reads it has just issued for the NEXT group)
  __amdgpu_buffer_rsrc_t i_wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W1), 0, 0x7ffffffe, 0x00020000);
  const __amdgpu_buffer_rsrc_t i_ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.A), 0, 0x7ffffffe, 0x00020000);
  int i_wk = 0, i_ak = 0;                            // scalar byte offsets of the staged K-step inside W / A rows
  unsigned char* i_st = nullptr;
  unsigned i_soff = 0, i_stride = 0, i_soffA[2] = {0, 0};
  // Every step issues its 8 W pieces (and, when the step it stages belongs to a first product, 2 A pieces) without a
  // branch: control flow between the fragment reads and the MFMAs makes hipcc wait lgkmcnt(0) in front of every MFMA
  // group, which serialises read latency and MFMAs.  Whether the staged step has an A image is known at the call site
  // (it is two steps ahead of the consumer).  Once the producer has run out of tiles it keeps re-staging step 0 of the
  // last tile into the slot nobody reads any more (two steps per workgroup).
  auto begin_issue = [&](const bool has_a) {
    const bool live = c_tile < ntiles;
    const int cv = live ? c_v : 0;
    const bool first = cv < nt1;
    const int kk = first ? cv : cv - nt1;
    const unsigned Kc = first ? (unsigned)p.K1 : 512u;
    const unsigned short* wsel = first ? p.W1 : p.W2p;
    i_wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wsel), 0, 0x7ffffffe, 0x00020000);
    i_wk = kk * 64;
    i_ak = (first ? kk : 0) * 64;
    i_stride = 32u * Kc;                               // 16 rows of Kc bf16
    i_st = smem + c_slot * SB_STAGE;
    const int ln = sb_lane();
    const int schunk = (ln & 3) ^ gsw((ln >> 4) & 3);
    i_soff = ((unsigned)(wave * 128 + (ln >> 2)) * Kc + schunk * 8) * 2u;
    if (has_a) {
      const int tl = live ? c_tile : 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int m = tl * 128 + (wave * 2 + i) * 16 + (ln >> 2);
        m = m < p.M ? m : p.M - 1;
        i_soffA[i] = ((unsigned)m * (unsigned)p.lda + schunk * 8) * 2u;
      }
    }
    c_slot = c_slot == 2 ? 0 : c_slot + 1;
    const int nv = c_v + 1;
    const bool wrap = nv == S;
    c_v = live ? (wrap ? 0 : nv) : c_v;
    c_tile = live && wrap ? c_tile + G : c_tile;
  };
  auto issue_one = [&](const int g, const bool has_a) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(i_wr, (lds_ptr_t)(i_st + (wave * 8 + g) * 1024), 16, (int)i_soff, i_wk + g * (int)i_stride, 0, 0);
    if (has_a && g < 2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(i_ar, (lds_ptr_t)(i_st + SB_W + (wave * 2 + g) * 1024), 16, (int)i_soffA[g], i_ak, 0, 0);
  };
  auto end_wait = [&](const bool has_a, const int extra) {
    // retires the images issued one step ago (read in the next step), leaves this step's in flight; `extra` = register
    // loads issued between the two (the residual prefetch), which stay in flight too
    if (extra == 0) {
      if (has_a) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      if (has_a) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");          // extra == 16: one row block's residual
      else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
  };

  // ---- tables into LDS
  float* tab1 = reinterpret_cast<float*>(smem + SB_TAB1);
  float* tab2 = reinterpret_cast<float*>(smem + SB_TAB2);
  for (int k = tid; k < p.n_classes * 256; k += 256)
    *reinterpret_cast<f32x4*>(tab1 + 4 * k) = *reinterpret_cast<const f32x4*>(p.fold + 4 * k);
  if (tid < 128) *reinterpret_cast<f32x4*>(tab2 + 4 * tid) = *reinterpret_cast<const f32x4*>(p.gamma + 4 * tid);
  else *reinterpret_cast<f32x4*>(tab2 + 4 * tid) = *reinterpret_cast<const f32x4*>(p.beta + 4 * (tid - 128));
  // biases and the clips' style classes too: with one wave per SIMD nothing hides a global-memory round trip in an
  // epilogue, so everything the epilogues read besides the residual rows comes from LDS
  float* lbias = reinterpret_cast<float*>(smem + SB_BIAS);
  for (int k = tid; k < 512; k += 256) {
    lbias[k] = p.b1[k];
    lbias[512 + k] = p.b2[k];
  }
  unsigned char* lcls = smem + SB_CLS;
  const bool cls_in_lds = p.n_label <= SB_MAXLAB;
  if (cls_in_lds)
    for (int k = tid; k < p.n_label; k += 256) {
      const long long lb = p.label[k];
      lcls[k] = (unsigned char)(lb < 0 ? 0 : (lb >= p.n_classes ? p.n_classes - 1 : (int)lb));
    }

  // ---- prologue: K-steps 0 and 1 of the first tile
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    begin_issue(true);
#pragma unroll
    for (int g = 0; g < 8; ++g) issue_one(g, true);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int slot = 0;                                                  // consumer ring slot

  f32x4 acc[2][32];
  bf16x8 hf[2][16];                                              // rows as the next product's B operand
  const bf16x8 zf = {};

  // one K-step: barrier, stage the step two ahead (one DMA per MFMA group), 64 MFMAs in 8 groups of 4 W fragments x 2
  // row blocks; the fragments of group g + 1 are read while group g multiplies (two named register sets, the scheduling
  // fences keep hipcc from serialising read -> wait -> MFMA), counted wait at the end
#define SB_FENCE()                         \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
  // Carried from one step to the next (also across the epilogues): the first four W fragments of the upcoming step and,
  // when it is a step of a first product, its two A fragments.  They are read in the previous step's last MFMA group:
  // with one wave per SIMD nothing else covers the barrier, the address arithmetic and the first read latency of a step.
  bf16x8 wA[4], nA0, nA1;
  auto kstep = [&](const bf16x8& b0, const bf16x8& b1, const bool from_lds, const bool has_a, const int extra, const bool next_from_lds) {
    begin_issue(has_a);
    const int fl = sb_lane(), fr = fl & 15, fq = fl >> 4;
    const int foff = fr * 64 + ((fq ^ gsw((fr >> 2) & 3)) << 4);   // fragment read offset inside a 16-row block
    const unsigned char* sb = smem + slot * SB_STAGE + foff;
    slot = slot == 2 ? 0 : slot + 1;
    const unsigned char* sbn = smem + slot * SB_STAGE + foff;      // the next step's image
    const bf16x8 a0 = from_lds ? nA0 : b0, a1 = from_lds ? nA1 : b1;
    bf16x8 wB[4];
    SB_FENCE();
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g < 7) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(sb + ((g + 1) * 4 + j) * 1024);
          if (g & 1) wA[j] = v; else wB[j] = v;
        }
      }
      issue_one(g, has_a);
      if (g == 7) {
        // this step's pieces are all issued: retire the next step's image (issued one step ago), meet the other waves,
        // and fetch its first fragments under this step's last eight MFMAs
        end_wait(has_a, extra);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) wA[j] = *reinterpret_cast<const bf16x8*>(sbn + j * 1024);
        if (next_from_lds) {
          nA0 = *reinterpret_cast<const bf16x8*>(sbn + SB_W + (wave * 2 + 0) * 1024);
          nA1 = *reinterpret_cast<const bf16x8*>(sbn + SB_W + (wave * 2 + 1) * 1024);
        }
      }
      SB_FENCE();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nb = g * 4 + j;
        const bf16x8 wf = (g & 1) ? wB[j] : wA[j];
        acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, a0, acc[0][nb], 0, 0, 0);
        acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, a1, acc[1][nb], 0, 0, 0);
      }
      SB_FENCE();
    }
  };
  // The 2 x 32 accumulators live in the AGPR half of the register file (256 registers); everything the vector ALU
  // touches lives in the other half.  hipcc would otherwise copy a whole row block (128 registers) out for the
  // LayerNorm passes, keep it there and spill the fragments: the pins below end every accumulator's VGPR live range
  // between passes, so each pass re-reads what it needs (v_accvgpr_read, one VALU slot each).
#define SB_PIN4(mb, nb) asm volatile("" : "+a"(acc[mb][nb]), "+a"(acc[mb][(nb) + 1]), "+a"(acc[mb][(nb) + 2]), "+a"(acc[mb][(nb) + 3]))
#define SB_PIN_MB(mb)                       \
  do {                                      \
    SB_PIN4(mb, 0);  SB_PIN4(mb, 4);  SB_PIN4(mb, 8);  SB_PIN4(mb, 12); \
    SB_PIN4(mb, 16); SB_PIN4(mb, 20); SB_PIN4(mb, 24); SB_PIN4(mb, 28); \
  } while (0)
  auto init_acc = [&](const float* bias) {   // bias: LDS
    const int iq = sb_lane() >> 4;
#pragma unroll
    for (int nb = 0; nb < 32; nb += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + (nb + j) * 16 + 4 * iq);
        acc[0][nb + j] = bv;
        acc[1][nb + j] = bv;
      }
      SB_PIN4(0, nb);
      SB_PIN4(1, nb);
    }
  };
  // two bf16 row pieces (accumulators nb, nb + 1 of one row block) -> 16 bytes of 8 consecutive columns per lane:
  // the accumulators exchange 16-lane rows (v_permlane16_swap); column 16 (nb + (q & 1)) + 8 (q >> 1)
  auto pair_cols = [&](const f32x4& lo, const f32x4& hi) {
    const unsigned x0 = sb_pack(lo[0], lo[1]), x1 = sb_pack(lo[2], lo[3]);
    const unsigned y0 = sb_pack(hi[0], hi[1]), y1 = sb_pack(hi[2], hi[3]);
    const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
    return u32x4{s0[0], s1[0], s0[1], s1[1]};
  };
  // LayerNorm statistics of one row block straight from the accumulators: mean and 1/std per row
  auto row_stats = [&](auto& A, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < 32; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += A[nb][e];
    s = row_lanes_sum(s);
    mean = s * (1.f / 512.f);
    float v = 0.f;
#pragma unroll
    for (int nb = 0; nb < 32; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = A[nb][e] - mean;
        v += d * d;
      }
    v = row_lanes_sum(v);
    rstd = __builtin_amdgcn_rsqf(v * (1.f / 512.f) + 1e-5f);
  };

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the table writes above
  __builtin_amdgcn_s_barrier();
  {
    const int fl = sb_lane(), fr = fl & 15, fq = fl >> 4;
    const unsigned char* sb0 = smem + fr * 64 + ((fq ^ gsw((fr >> 2) & 3)) << 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) wA[j] = *reinterpret_cast<const bf16x8*>(sb0 + j * 1024);
    nA0 = *reinterpret_cast<const bf16x8*>(sb0 + SB_W + (wave * 2 + 0) * 1024);
    nA1 = *reinterpret_cast<const bf16x8*>(sb0 + SB_W + (wave * 2 + 1) * 1024);
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += G) {
    const int m_base = tile * 128 + wave * 32;
    // ---- product 1: t = A . W1^T + b1, A and W1 through the ring
    init_acc(lbias);
    for (int t = 0; t < nt1 - 2; ++t) kstep(zf, zf, true, true, 0, true);   // stages steps 2 .. nt1-1 of this product
    kstep(zf, zf, true, false, 0, true);                                      // stage the second product's first two steps
    kstep(zf, zf, true, false, 0, false);
    // ---- stylization pre-op in registers: h = SiLU(LN(t) gamma'_c + beta'_c) -> bf16 B-operand fragments
    {
      const int el = sb_lane(), er = el & 15, eq = el >> 4;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int m = m_base + mb * 16 + er;
        const int mc = m < p.M ? m : p.M - 1;
        const int clip = p.n_label == 1 ? 0 : mc / p.rows_per_clip;
        int cls;
        if (cls_in_lds) {
          cls = lcls[clip];
        } else {
          const long long lb = p.label[clip];
          cls = lb < 0 ? 0 : (lb >= p.n_classes ? p.n_classes - 1 : (int)lb);
        }
        // one pass over the accumulators for both moments (every v_accvgpr_read is a VALU slot of a lone wave)
        float mean, rstd;
        {
          // two passes (mean, then squared deviations), as the two-launch path's row kernel and nn.LayerNorm do: the
          // one-pass E[x^2] - mean^2 loses the variance of rows whose mean is large against their spread, and which
          // of the two paths a tail takes depends on the batch size
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int nb = 0; nb < 32; ++nb)
#pragma unroll
            for (int e = 0; e < 4; ++e) s1 += acc[mb][nb][e];
          mean = row_lanes_sum(s1) * (1.f / 512.f);
#pragma unroll
          for (int nb = 0; nb < 32; ++nb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = acc[mb][nb][e] - mean;
              s2 = __builtin_fmaf(d, d, s2);
            }
          rstd = __builtin_amdgcn_rsqf(row_lanes_sum(s2) * (1.f / 512.f) + 1e-5f);
        }
        SB_PIN_MB(mb);
        const float nmr = -mean * rstd;
        const float* gp = tab1 + cls * 1024 + 4 * eq;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
          unsigned pk[4];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int nb = 2 * kb + h2;
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gp + nb * 16);
            const f32x4 be = *reinterpret_cast<const f32x4*>(gp + 512 + nb * 16);
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              y[e] = __builtin_fmaf(__builtin_fmaf(acc[mb][nb][e], rstd, nmr), ga[e], be[e]);
              // SiLU = y / (1 + 2^(-y log2 e)) on the raw v_exp_f32 / v_rcp_f32
              y[e] = y[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(y[e] * -1.4426950408889634f));
            }
            pk[2 * h2] = sb_pack(y[0], y[1]);
            pk[2 * h2 + 1] = sb_pack(y[2], y[3]);
          }
          const u32x4 v = {pk[0], pk[1], pk[2], pk[3]};
          hf[mb][kb] = __builtin_bit_cast(bf16x8, v);
          asm volatile("" : "+v"(hf[mb][kb]));       // computed HERE: MachineSink would move the arithmetic to the K-step that reads it
          if ((kb & 1) == 1) SB_FENCE();
        }
      }
    }
    // ---- product 2: h . W2^T + b2, rows from registers
    init_acc(lbias + 512);
    // the residual rows x[m][:] of the FIRST row block are fetched into registers during the last steps (the fragments of
    // the consumed k-blocks have freed them); the second block's are issued in the epilogue once the first block's have
    // been added, and land while the first block is normalised and stored.  Both at once (128 VGPRs) do not fit beside the
    // live fragments: the allocator spills them with a vmcnt(0) each.  A lane reads the 8 consecutive bf16 it will
    // overwrite (store layout), 16 x 16 B per row block.
    u32x4 xr[16];
    const int xl = sb_lane(), xer = xl & 15, xeq = xl >> 4;
    constexpr int XPF = 12;                                        // step after which the prefetch is issued
    auto fetch_residual = [&](int mb) {
      // rows past M read row M - 1 (never stored): a select on the loaded value would be a USE, and the wait for it
      // would sit at the load instead of in the epilogue
      int m = m_base + mb * 16 + xer;
      m = m < p.M ? m : p.M - 1;
      const unsigned short* xrow = p.X + (size_t)m * 512 + 16 * (xeq & 1) + 8 * (xeq >> 1);
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) xr[jj] = *reinterpret_cast<const u32x4*>(xrow + 32 * jj);
      SB_FENCE();
    };
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      kstep(hf[0][kb], hf[1][kb], false, kb >= 14, kb == XPF + 1 ? 16 : 0, kb == 15);
      if (kb == XPF) fetch_residual(0);
    }
    // ---- residual + post-LN: x_new = LN(x + .), written back in place (bf16 residual stream)
    {
      const int el = sb_lane(), er = el & 15, eq = el >> 4;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int m = m_base + mb * 16 + er;
        const bool live = m < p.M;
        unsigned short* xrow = p.X + (size_t)(live ? m : 0) * 512 + 16 * (eq & 1) + 8 * (eq >> 1);
        // residual add, un-paired with the same v_permlane16_swap the stores pair with; both moments on the way
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const auto a0 = __builtin_amdgcn_permlane16_swap(xr[jj][0], xr[jj][2], false, false);
          const auto a1 = __builtin_amdgcn_permlane16_swap(xr[jj][1], xr[jj][3], false, false);
          f32x4 lo = acc[mb][2 * jj], hi = acc[mb][2 * jj + 1];
          lo[0] += __builtin_bit_cast(float, a0[0] << 16);
          lo[1] += __builtin_bit_cast(float, a0[0] & 0xffff0000u);
          lo[2] += __builtin_bit_cast(float, a1[0] << 16);
          lo[3] += __builtin_bit_cast(float, a1[0] & 0xffff0000u);
          hi[0] += __builtin_bit_cast(float, a0[1] << 16);
          hi[1] += __builtin_bit_cast(float, a0[1] & 0xffff0000u);
          hi[2] += __builtin_bit_cast(float, a1[1] << 16);
          hi[3] += __builtin_bit_cast(float, a1[1] & 0xffff0000u);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s1 += lo[e] + hi[e];
            s2 = __builtin_fmaf(lo[e], lo[e], s2);
            s2 = __builtin_fmaf(hi[e], hi[e], s2);
          }
          acc[mb][2 * jj] = lo;
          acc[mb][2 * jj + 1] = hi;
          if ((jj & 1) == 1) {
            SB_PIN4(mb, 2 * jj - 2);
            SB_FENCE();
          }
        }
        if (mb == 0) fetch_residual(1);
        s1 = row_lanes_sum(s1);
        s2 = row_lanes_sum(s2);
        const float mean = s1 * (1.f / 512.f);
        const float var = __builtin_fmaf(-mean, mean, s2 * (1.f / 512.f));
        const float rstd = __builtin_amdgcn_rsqf((var > 0.f ? var : 0.f) + 1e-5f);
        const float nmr = -mean * rstd;
        const float* gp = tab2 + 4 * eq;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
          f32x4 y[2];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int nb = 2 * kb + h2;
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gp + nb * 16);
            const f32x4 be = *reinterpret_cast<const f32x4*>(gp + 512 + nb * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[h2][e] = __builtin_fmaf(__builtin_fmaf(acc[mb][nb][e], rstd, nmr), ga[e], be[e]);
          }
          const u32x4 o = pair_cols(y[0], y[1]);
          if (live) *reinterpret_cast<u32x4*>(xrow + 32 * kb) = o;
          if ((kb & 1) == 1) SB_FENCE();
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// K-axis permutation of a [N][K] bf16 weight for products that take their rows from accumulators (see the header):
// out[n][32 kb + 8 q + j] = in[n][32 kb + (j < 4 ? 4 q + j : 16 + 4 q + (j - 4))]
__global__ void permute_k32_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, size_t total, int K) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(e % K);
    const int kb = k >> 5, s = k & 31, qq = s >> 3, j = s & 7;
    const int f = j < 4 ? 4 * qq + j : 16 + 4 * qq + (j - 4);
    out[e] = in[e - k + kb * 32 + f];
  }
}
int launch_permute_k32(const unsigned short* in, unsigned short* out, int N, int K, hipStream_t s) {
  PT_CHECK(in && out && N > 0 && K > 0 && K % 32 == 0, "permute_k32: K=%d must be a positive multiple of 32", K);
  const size_t total = (size_t)N * K;
  PT_LAUNCH(permute_k32_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, in, out, total, K);
  return 0;
}

int launch_sublayer(const SubArgs& a, hipStream_t s) {
  PT_CHECK(a.M > 0 && a.A && a.W1 && a.b1 && a.fold && a.label && a.W2p && a.b2 && a.gamma && a.beta && a.X, "sublayer: null argument");
  PT_CHECK(a.K1 >= 64 && a.K1 % 32 == 0 && a.lda >= a.K1 && a.lda % 8 == 0, "sublayer: K1=%d (multiple of 32, >= 64) lda=%d", a.K1, a.lda);
  PT_CHECK(a.n_classes >= 1 && a.n_classes <= 4, "sublayer: %d style classes (the LDS tables hold 4)", a.n_classes);
  PT_CHECK(a.n_label == 1 || (a.rows_per_clip > 0 && (long long)a.n_label * a.rows_per_clip >= a.M), "sublayer: %d labels x %d rows < M=%d",
           a.n_label, a.rows_per_clip, a.M);
  PT_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W1 & 15) == 0 && ((uintptr_t)a.W2p & 15) == 0 && ((uintptr_t)a.X & 15) == 0,
           "sublayer: operands must be 16-byte aligned");
  PT_CHECK((double)a.M * a.lda * 2 < 2.0e9, "sublayer: A larger than 2 GB (32-bit buffer offsets)");
  const int sb_num_cus = device_cus();
  PT_ENSURE_LDS(sublayer_kernel, SB_SMEM);
  const int ntiles = (a.M + 127) / 128;
  const int grid = ntiles < sb_num_cus ? ntiles : sb_num_cus;
  void* tok = nullptr;
  if (prof_is_on()) prof_record_begin(s, 2.0 * a.M * 512.0 * (a.K1 + 512.0), 7, &tok);
  PT_LAUNCH(sublayer_kernel, dim3(grid), dim3(256), SB_SMEM, s, a, ntiles);
  if (tok) prof_record_end(s, tok);
  return 0;
}

// Workgroups the persistent grid runs with on this device (the dispatch in transformer.hip prices whole rounds of it).
int sublayer_grid_limit() {
  return device_cus();
}

}  // namespace pt
