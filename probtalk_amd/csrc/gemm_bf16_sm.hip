// Small-batch bf16 GEMM (round 3, the latency regime: 1-16 clips, a few hundred to ~2,400 rows per Linear).
//
// At these sizes a launch of the persistent kernels is ONE round of tiles on a fraction of the CUs, and its length is the
// K loop of a single tile paced by the depth of its staging ring: gemm_row_kernel<2,.> streams the whole W (512 KB at
// K = 512) through each of its 44 workgroups three 32-deep K-tiles at a time, 0.75 us per K-tile whatever the MFMAs do.
// Here a workgroup owns a 64 x 64 output tile, so 1,408 x 512 is 176 workgroups instead of 44, its operands for the
// whole of K <= 512 (64 + 64 rows x K x 2 B = 128 KB) are requested by LDS-DMA before the first MFMA and the K loop follows the
// arrival of the data: one memory round trip per launch instead of one per two K-tiles.  Longer K runs through the same
// 128 KB as a ring of four 128-deep groups.
//
//   512 threads = 8 waves as 2 (M) x 4 (N), wave tile 32 x 16 = 2 v_mfma_f32_16x16x32_bf16 accumulators (operands
//   swapped as in the row kernel: a lane owns one row and 4 consecutive columns);
//   LDS image of a 32-deep K-tile: [A: 64 rows | W: 64 rows] x 64 B in the row kernel's format (16-row pieces of 1 KB,
//   one DMA instruction each, chunk swizzle in the source address); waves 0-3 issue the A pieces, waves 4-7 the W
//   pieces: 1 DMA instruction per wave per K-tile, 4 per group (with 4 waves issuing 2 each a launch was 0.3-0.9 us longer);
//   group g is waited for with vmcnt(4 x groups issued after it), then s_barrier; a ring slot is re-filled after a
//   second barrier (every wave has read it).
// Epilogue: + bias, ReLU, fp32 and / or bf16 rows straight from the accumulators.  No positional table, residual or
// zero-padded columns (those Linears keep their kernels).  LayerNorm does not fuse here (a row spans 8 workgroups): the
// small-batch path runs the row kernels of the fp32 mode behind it (transformer.hip).
#include <cstdlib>

#include "pt_common.h"

namespace pt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

void prof_record_begin(hipStream_t s, double flops, int cfg, void** tok);
void prof_record_end(hipStream_t s, void* tok);
bool prof_is_on();

constexpr int SM_KT = 8192;               // bytes of one 32-deep K-tile image: A 4 KB | W 4 KB
constexpr int SM_GROUP = 4;               // K-tiles per group (128 of K)
constexpr int SM_RING = 4;                // groups in the ring
constexpr int SM_SMEM = SM_RING * SM_GROUP * SM_KT;

__global__ __launch_bounds__(512) void gemm_sm_kernel(const GemmBigArgs p, const int tiles_n) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;                   // 8 waves as 2 (M) x 4 (N): wave tile 32 x 16
  const int r = lane & 15, q4 = lane >> 4;
  const int ng = p.K >> 7;                                   // groups of 128 (launcher: K % 128 == 0)
  auto gsw = [](int k) { return (0x78 >> (2 * k)) & 3; };    // g = {0, 2, 3, 1}: the row kernel's chunk swizzle

  // bias of the lane's 2 x 4 columns (used in the epilogue only)
  f32x4 bv;
  {
    const int n = tn * 64 + wc * 16 + 4 * q4;
    bv = (p.bias && n < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // DMA sources: waves 0-3 issue piece `wave` of A, waves 4-7 piece `wave - 4` of W (image offset wave * 1 KB either
  // way); lane -> row (lane >> 2) of the piece, LDS chunk (lane & 3)
  const unsigned char* src;
  {
    const int lr = (wave & 3) * 16 + (lane >> 2);
    const int g = (lane & 3) ^ gsw((lr >> 2) & 3);
    if (wave < 4) {
      int m = tm * 64 + lr;
      m = m < p.M ? m : p.M - 1;
      src = reinterpret_cast<const unsigned char*>(p.A) + ((size_t)m * p.lda + g * 8) * 2u;
    } else {
      int n = tn * 64 + lr;
      n = n < p.Npad ? n : p.Npad - 1;
      src = reinterpret_cast<const unsigned char*>(p.W) + ((size_t)n * p.K + g * 8) * 2u;
    }
  }
  auto issue_group = [&](const int g) {
    unsigned char* dst = smem + (g & (SM_RING - 1)) * (SM_GROUP * SM_KT) + wave * 1024;
#pragma unroll
    for (int t = 0; t < SM_GROUP; ++t)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)(g * SM_GROUP + t) * 64), (lds_ptr_t)(dst + t * SM_KT), 16, 0, 0);
  };
  const int pre = ng < SM_RING ? ng : SM_RING;
  for (int g = 0; g < pre; ++g) issue_group(g);

  f32x4 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int foff = r * 64 + ((q4 ^ gsw((r >> 2) & 3)) << 4);  // fragment read offset inside a 16-row piece

  for (int g = 0; g < ng; ++g) {
    // groups issued after g: min(ng, g + SM_RING) - (g + 1), 4 DMA instructions each
    const int younger = (ng < g + SM_RING ? ng : g + SM_RING) - (g + 1);
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* gb = smem + (g & (SM_RING - 1)) * (SM_GROUP * SM_KT);
#pragma unroll
    for (int t = 0; t < SM_GROUP; ++t) {
      bf16x8 a[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) a[j] = *reinterpret_cast<const bf16x8*>(gb + t * SM_KT + (wr * 2 + j) * 1024 + foff);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(gb + t * SM_KT + 4096 + wc * 1024 + foff);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a[j], acc[j], 0, 0, 0);
    }
    if (g + SM_RING < ng) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's fragment reads of the slot are done
      __builtin_amdgcn_s_barrier();                          // ... and every other wave's
      issue_group(g + SM_RING);
    }
  }

  // acc[j][e]: row m = tm*64 + wr*32 + j*16 + r, column n = tn*64 + wc*16 + 4*q4 + e
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = tm * 64 + wr * 32 + j * 16 + r;
    if (m >= p.M) continue;
    {
      const int n = tn * 64 + wc * 16 + 4 * q4;
      if (n >= p.N) continue;
      f32x4 v = acc[j] + bv;
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      if (p.C) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
      if (p.Cb) {
        const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        *reinterpret_cast<u32x2*>(p.Cb + (size_t)m * p.ldcb + n) = __builtin_bit_cast(u32x2, h);
      }
    }
  }
}

// EPI_STD launches this kernel can take (launch_gemm_big asks before it routes a small launch here)
bool gemm_sm_takes(const GemmBigArgs& a) {
  return a.epi == EPI_STD && !a.pe && !a.Res && a.zero_to == 0 && a.K >= 128 && a.K % 128 == 0 && a.N % 4 == 0 &&
         (!a.C || (a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0)) && (!a.Cb || (a.ldcb % 4 == 0 && ((uintptr_t)a.Cb & 7) == 0)) &&
         (!a.bias || ((uintptr_t)a.bias & 15) == 0);
}

// ... and wants: few enough tiles that its one-round latency beats the persistent kernels (measured, tools/micro/sm_bench.py:
// N = 512 up to ~3,000 rows against gemm_row_kernel<2,.>, wide Linears up to ~768 tiles against the 4-wave kernel)
bool gemm_sm_wants(const GemmBigArgs& a) {
  if (a.M > option(OPT_SM_ROWS) || !gemm_sm_takes(a)) return false;
  const long tiles = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
  return tiles <= (a.N == 512 ? 384 : 768);
}

int launch_gemm_sm(const GemmBigArgs& a, hipStream_t s) {
  PT_CHECK(gemm_sm_takes(a), "gemm_sm: launch not supported (internal dispatch error)");
  PT_ENSURE_LDS(gemm_sm_kernel, SM_SMEM);
  const int tiles_m = (a.M + 63) / 64, tiles_n = (a.N + 63) / 64;
  void* tok = nullptr;
  if (prof_is_on()) prof_record_begin(s, a.flops > 0 ? a.flops : 2.0 * a.M * a.N * a.K, 5, &tok);
  PT_LAUNCH(gemm_sm_kernel, dim3(tiles_m * tiles_n), dim3(512), SM_SMEM, s, a, tiles_n);
  if (tok) prof_record_end(s, tok);
  return 0;
}

}  // namespace pt
