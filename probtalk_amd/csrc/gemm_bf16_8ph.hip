// Phase-pipelined 256 x 256 bf16 GEMM for the wide transformer Linears (in-proj 512->1536, FFN-1 512->2048,
// cross K/V 512->1024, heads): C[M,N] = A[M,K] . W[N,K]^T, bf16 operands, fp32 accumulation.
//
// 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 v_mfma_f32_16x16x32_bf16 accumulators.
// A K-tile (BK = 64) is staged by LDS-DMA (global_load_lds_dwordx4) as FOUR half-tile images of 128 rows x 128 B:
//   h0 / h1 = the 64-row halves mh = 0 / 1 of every wave's A rows,  h2 / h3 = the 32-column halves nh = 0 / 1 of every
//   wave's W rows, two K-tile buffers (2 x 4 x 16 KB = 128 KB).
// A K-tile is four phases of 16 MFMAs, one output quadrant (mh, nh) x K = 64 each:
//   P1 reads B(nh0) + A(mh0), MFMA (0,0);  P2 reads B(nh1), MFMA (0,1);  P3 reads A(mh1), MFMA (1,1);  P4 MFMA (1,0).
// Every phase is  [ds_reads | DMA issue of ONE half-tile | counted vmcnt]  s_barrier  [16 MFMAs]  s_barrier, and
// waves 4-7 run one barrier behind waves 0-3: the two waves that share a SIMD alternate between the read/DMA segment
// and the MFMA segment, so the matrix pipe always has a wave to issue from (MI355X_MICROARCH.md, two waves per SIMD).
//
// DMA schedule (phase p of K-tile t): P1 -> h3 of t+1, P2 -> h1 of t+1, P3 -> h0 of t+2, P4 -> h2 of t+2: a half-tile
// is re-filled two phases or more after its last read (WAR), is retired by the counted wait four phases after its issue
// (vmcnt(8) = four half-tiles of 2 DMA instructions stay in flight) and is read five or six phases after its issue,
// i.e. at least one phase and one barrier after every wave's retiring wait (RAW across the staggered halves).
// The K-tile stream runs across output tiles (persistent workgroups): the next tile's first K-tiles land while the
// current tile's accumulators are stored.  Stores share the vmcnt counter with the DMA (in-order retirement), so the
// counted waits have to know how many stores sit between a half-tile and the wait:
//   * plain bf16 output, interior tile, a successor tile exists: the upper half of the tile (mh = 0, quadrants (0,0) and
//     (0,1)) is final after P2 of the last K-tile.  Its 8 store instructions per wave leave at the head of P3 and P4 (4
//     each: two 16-row blocks x both column halves, so every row's 128-byte line is written by two back-to-back
//     instructions), under those phases' MFMAs; the lower half (final after P4) follows the loop.  P4's wait counts the 4
//     stores in (vmcnt(12)) and the tile ends with vmcnt(20): the 16 stores and the two half-tiles of P3 / P4 may stay in
//     flight, the next tile's first K-tile (last issued in P2) has landed.  Store counts are exact because the tile is
//     interior (no lane or wave is predicated off).  A/B on one box at configs[1], over the 122 launches of this kernel
//     per step: 23.4 ms without, 21.8 ms storing each quadrant one phase after its last MFMA (half lines in different
//     phases), 20.7 ms this way;
//   * every other tile (edges, fp32 / positional / residual epilogues, the last tile of a workgroup) ends with vmcnt(0)
//     BEFORE its stores.
// Either way everything the next tile's first K-tile reads has landed by then.  After a vmcnt(0) that K-tile issues no
// waits; after an early tile its P3 / P4 retire the two half-tiles still in flight behind the stores (vmcnt(24) / (20)).
// The bias row sits in LDS (written once per workgroup): accumulators start at the bias, and a bias read from memory at the
// head of every tile made the compiler wait for vmcnt(0) there, i.e. for every store of the tile before and every DMA in
// flight (round 3, profiles/r3/tile_head_bias_loads.txt).
#include <cstdlib>

#include "pt_common.h"

namespace pt {

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

constexpr int G8_HT = 128 * 128;          // bytes of a half-tile image (128 rows x 64 bf16)
constexpr int G8_STAGE = 4 * G8_HT;       // one K-tile buffer
constexpr int G8_RING = 2 * G8_STAGE;
constexpr int G8_BIAS_COLS = 8192;        // bias row kept in LDS (fp32): columns of the widest launch that uses it
constexpr int G8_SMEM = G8_RING + G8_BIAS_COLS * 4;

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

__global__ __launch_bounds__(512) void gemm8_kernel(const GemmBigArgs p, const int tiles_n, const int ntiles, const int overlap) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int G = gridDim.x;
  int tile0;
  {
    const int q = G >> 3, rr = G & 7, x = blockIdx.x & 7;       // bijective XCD-aware order (see gemm_bf16_big.hip)
    tile0 = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (blockIdx.x >> 3);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, q4 = lane >> 4;
  const int nt = p.K >> 6;

  // ---- per-half-tile DMA cursors (wave-uniform) and per-lane source offsets
  int c_tile[4], c_kt[4], c_buf[4];
  unsigned soff[4][2];
  auto set_src = [&](int h, int tl) {
    const int tm = tl / tiles_n, tn = tl - tm * tiles_n;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (wave + 8 * i) * 8 + (lane >> 3);              // local row of the half-tile image
      const int g = (lane & 7) ^ ((lr >> 1) & 7);                   // source chunk (swizzle lives in the source address)
      if (h < 2) {
        int m = tm * 256 + (lr >> 6) * 128 + h * 64 + (lr & 63);
        m = m < p.M ? m : p.M - 1;
        soff[h][i] = ((unsigned)m * (unsigned)p.lda + g * 8) * 2u;
      } else {
        int n = tn * 256 + (lr >> 5) * 64 + (h - 2) * 32 + (lr & 31);
        n = n < p.Npad ? n : p.Npad - 1;
        soff[h][i] = ((unsigned)n * (unsigned)p.K + g * 8) * 2u;
      }
    }
  };
  bool draining = false;
  auto issue = [&](int h) {
    if (c_tile[h] >= ntiles) {
      draining = true;
      return;
    }
    const unsigned char* base = reinterpret_cast<const unsigned char*>(h < 2 ? (const void*)p.A : (const void*)p.W) + (size_t)c_kt[h] * 128;
    unsigned char* dst = smem + c_buf[h] * G8_STAGE + h * G8_HT + wave * 1024;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + soff[h][0]), (lds_ptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + soff[h][1]), (lds_ptr_t)(dst + 8 * 1024), 16, 0, 0);
    c_buf[h] ^= 1;
    if (++c_kt[h] == nt) {
      c_kt[h] = 0;
      c_tile[h] += G;
      if (c_tile[h] < ntiles) set_src(h, c_tile[h]);
    }
  };
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    c_tile[h] = tile0;
    c_kt[h] = 0;
    c_buf[h] = 0;
    set_src(h, tile0);
  }
  // prologue: K-tile 0 whole, then h0 / h2 of K-tile 1 (the loop issues h3 / h1 of t+1 and h0 / h2 of t+2)
  issue(0); issue(2); issue(3); issue(1); issue(0); issue(2);
  float* bias_s = reinterpret_cast<float*>(smem + G8_RING);
  for (int n = tid * 4; n < tiles_n * 256; n += 2048) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + n);      // N % 8 == 0 (launcher)
    *reinterpret_cast<f32x4*>(bias_s + n) = bv;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();                    // waves 4-7 run one barrier behind

  // fragment read offsets inside a half-tile image: row r, chunk (4 s + q4) ^ ((r >> 1) & 7)
  int aoff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) aoff[s] = r * 128 + (((4 * s + q4) ^ ((r >> 1) & 7)) << 4);
  const int a_row0 = wr * 64 * 128, b_row0 = wc * 32 * 128;

  f32x4 acc[8][4];
  bf16x8 af[4][2], bfr[2][2][2];
  int cbuf = 0;

  bool after_early = false;
  for (int tile = tile0; tile < ntiles; tile += G) {
    {
      // accumulators start at the bias (its loads overlap the first K-tile instead of standing in the epilogue)
      int iq = q4;
      asm volatile("" : "+v"(iq));
      const int tn0 = tile - (tile / tiles_n) * tiles_n;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = tn0 * 256 + wc * 64 + j * 16 + 4 * iq;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + n);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][j] = bv;
      }
    }

    // An interior tile with a successor and a plain bf16 output stores its upper half (mh = 0, final after P2 of the last
    // K-tile) under the MFMAs of P3 / P4 instead of after the loop (see the header)
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const bool early = overlap && p.Cb && !p.C && !p.pe && !p.Res && tile + G < ntiles && tm * 256 + 256 <= p.M &&
                       tn * 256 + 256 <= p.N;
    // n 16-row blocks of the half mh, both column halves of each: the wave's whole 128-byte line of a row leaves in two
    // back-to-back instructions (storing the column halves in different phases cost 13 % more HBM write bytes by PMC)
    auto storeRows = [&](int mh, int i0, int n) {
      int er = r, eq = q4;
      asm volatile("" : "+v"(er), "+v"(eq));
      unsigned short* cb = p.Cb + (size_t)(tm * 256 + wr * 128 + mh * 64 + er) * p.ldcb + tn * 256 + wc * 64 + 16 * (eq & 1) + 8 * (eq >> 1);
      for (int i = i0; i < i0 + n; ++i) {
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          f32x4 a0 = acc[mh * 4 + i][j], a1 = acc[mh * 4 + i][j + 1];
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a0[e] = a0[e] > 0.f ? a0[e] : 0.f;
              a1[e] = a1[e] > 0.f ? a1[e] : 0.f;
            }
          }
          const auto s0 = __builtin_amdgcn_permlane16_swap(pack_bf16(a0[0], a0[1]), pack_bf16(a1[0], a1[1]), false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(pack_bf16(a0[2], a0[3]), pack_bf16(a1[2], a1[3]), false, false);
          const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4*>(cb + (size_t)i * 16 * p.ldcb + 16 * j) = v;
        }
      }
    };

    for (int t = 0; t < nt; ++t, cbuf ^= 1) {
      const unsigned char* sb = smem + cbuf * G8_STAGE;
      const bool fl = early && t == nt - 1;
      auto readA = [&](int mh) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            af[i][s] = *reinterpret_cast<const bf16x8*>(sb + mh * G8_HT + a_row0 + i * 2048 + aoff[s]);
      };
      auto readB = [&](int nh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            bfr[nh][j][s] = *reinterpret_cast<const bf16x8*>(sb + (2 + nh) * G8_HT + b_row0 + j * 2048 + aoff[s]);
      };
      // `stores` = early store instructions of this K-tile already in the queue: they are younger than the half-tile
      // this wait retires, so the count grows by them and the DMA retires exactly where it does without them
      // First K-tile of a tile: after the prologue or a tile that ended with vmcnt(0) nothing is pending.  After an
      // `early` tile the half-tiles that tile issued in its last P3 / P4 (this tile's K-tile 1, h0 / h2) are still in
      // flight behind its stores; they are read in P1 of K-tile 1, so P3 / P4 here retire them as every other K-tile's
      // P3 / P4 would: younger than the P3 half-tile are 4 + 8 stores, the P4 half-tile and its 4 stores and this tile's
      // three half-tiles so far = 24 instructions; younger than the P4 half-tile 4 + 8 stores and four half-tiles = 20.
      auto wait = [&](int stores, int first = -1) {
        if (t > 0) {
          if (draining) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (fl && stores == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (after_early && first >= 0) {
          if (draining) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (first == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        }
      };
      auto mma = [&](int mh, int nh) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[mh * 4 + i][nh * 2 + j] =
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nh][j][s], af[i][s], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      };
      // P1
      readB(0); readA(0); issue(3); wait(0);
      mma(0, 0);
      // P2
      readB(1); issue(1); wait(0);
      mma(0, 1);
      // P3
      readA(1); issue(0); wait(0, 24);
      if (fl) storeRows(0, 0, 2);
      mma(1, 1);
      // P4
      issue(2); wait(4, 20);
      if (fl) storeRows(0, 2, 2);
      mma(1, 0);
    }
    if (early) {
      // the next tile's first K-tile issues no waits: its last half-tile (h1, issued in P2) must have landed.  Younger
      // than it: 16 stores and the two half-tiles of P3 / P4 = 20 instructions
      storeRows(1, 0, 4);
      asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      after_early = true;
      continue;
    }

    // ---- epilogue.  acc[i][j][e]: row m = m0 + wr*128 + i*16 + r, column n = n0 + wc*64 + j*16 + 4*q4 + e
    // (lane coordinates laundered so that the address arithmetic is not hoisted into the main loop's registers)
    int er = r, eq = q4;
    asm volatile("" : "+v"(er), "+v"(eq));
    const int mw = tm * 256 + wr * 128, nw = tn * 256 + wc * 64;
    const int ncols = p.N > p.zero_to ? p.N : p.zero_to;
    if (p.pe || p.Res || p.relu || nw + 64 > p.N) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nw + j * 16 + 4 * eq;
        const bool real = n < p.N;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = mw + i * 16 + er;
          const bool live = m < p.M;
          f32x4 v = acc[i][j];
          if (p.pe && real && live) {
            const f64x4 pv = *reinterpret_cast<const f64x4*>(p.pe + (size_t)(m % p.pe_period) * p.N + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (float)((double)v[e] + pv[e]);
          }
          if (p.Res && real && live) v += *reinterpret_cast<const f32x4*>(p.Res + (size_t)m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (p.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
            v[e] = real ? v[e] : 0.f;
          }
          acc[i][j] = v;
        }
      }
    }
    // everything staged so far has landed before the stores join the vmcnt queue (see the header)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    after_early = false;
    // fp32 rows leave straight from the accumulators (16 rows x 64 B per instruction); bf16: accumulators j and j+1
    // exchange 16-lane rows (v_permlane16_swap), then a lane holds 8 consecutive bf16 columns
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = mw + i * 16 + er;
      const bool live = m < p.M;
      if (p.C) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = nw + j * 16 + 4 * eq;
          if (live && n < ncols) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = acc[i][j];
        }
      }
      if (p.Cb) {
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const unsigned x0 = pack_bf16(acc[i][j][0], acc[i][j][1]), x1 = pack_bf16(acc[i][j][2], acc[i][j][3]);
          const unsigned y0 = pack_bf16(acc[i][j + 1][0], acc[i][j + 1][1]), y1 = pack_bf16(acc[i][j + 1][2], acc[i][j + 1][3]);
          const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
          const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
          const int n = nw + 16 * (j + (eq & 1)) + 8 * (eq >> 1);
          if (live && n < ncols) *reinterpret_cast<u32x4*>(p.Cb + (size_t)m * p.ldcb + n) = v;
        }
      }
      asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3])::"memory");
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();                    // balances the extra barrier waves 4-7 took in the prologue
}

int launch_gemm8(const GemmBigArgs& a, int cfg, hipStream_t s) {
  PT_ENSURE_LDS(gemm8_kernel, G8_SMEM);
  const int g8_num_cus = device_cus();
  const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256, ntiles = tiles_m * tiles_n;
  const int grid = ntiles < g8_num_cus ? ntiles : g8_num_cus;
  void* tok = nullptr;
  if (prof_is_on()) prof_record_begin(s, a.flops > 0 ? a.flops : 2.0 * a.M * a.N * a.K, cfg, &tok);
  PT_LAUNCH(gemm8_kernel, dim3(grid), dim3(512), G8_SMEM, s, a, tiles_n, ntiles, option(OPT_STORE_OVERLAP));
  if (tok) prof_record_end(s, tok);
  return 0;
}

}  // namespace pt
