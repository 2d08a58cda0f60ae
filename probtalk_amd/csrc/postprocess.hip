// Pose post-process behind the generation path (scripts/demo.py:183-205), one kernel:
//   denormalise   x = pred * std + mean, statistics gathered per channel as nets/utils.py:16-43 ('fbhe': jaw 0..5 |
//                 c_index body+hands | the last 100 expression entries of the 430-entry statistics)
//   6-D -> matrix Gram-Schmidt (data_utils/rotation_conversion.py:512-533, F.normalize with eps 1e-12)
//   -> quaternion (rotation_conversion.py:97-119) -> axis-angle (:481-509, |angle| < 1e-6 branch)
//   part2full     the 27 fixed lower-body scalars inserted (data_utils/lower_body.py:66-83)
// pred is channels-first (B, 376, T) as the networks return it; the result is (B, T, 265) rows (238 without part2full).
//
// HBM-bound: 1,504 B read and 1,060 B written per frame.  A workgroup owns 32 frames of one clip: lanes run along t, so a
// channel's 32 values are one 128-byte segment; the axis-angle triples go to an LDS image of the 32 output rows, which are
// contiguous in memory (32 x 265 floats) and leave as full lines.
// The arithmetic follows torch's CPU kernels operation for operation, contraction off and fused multiply-adds exactly
// where torch has them (its vector norm accumulates fma(x, x, acc); its cross product is fma(a, b, -(c d)); sums and
// the rest round separately; checked bit for bit against torch in numpy fp32).  Gram-Schmidt on nearly parallel a1, a2 and
// sqrt(1 + trace) near angle = pi are cancelling sums: one differently rounded operation upstream moves the result by
// 1e-4, far beyond the 1e-5 parity bound, so "same formula" is not enough here.
#include "pt_common.h"

#pragma clang fp contract(off)

namespace pt {

constexpr int PP_C = 376, PP_J = 46, PP_E = 100, PP_TT = 32;

__device__ __forceinline__ int pp_out_col(int k, bool full) {
  // column of axis-angle scalar k (0..137) in the output row: part2full keeps [0,3) | lp 15 | [3,6) | lp 6 | [6,9) | lp 6 | rest
  if (!full) return k;
  return k < 3 ? k : (k < 6 ? k + 15 : (k < 9 ? k + 21 : k + 27));
}

__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ pred, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, const int* __restrict__ cidx,
                                                          int n_stats, const float* __restrict__ lower, float* __restrict__ out,
                                                          int T, int ld) {
  __shared__ float tile[PP_TT * 265];
  const bool full = lower != nullptr;
  const int W = full ? 265 : 238;
  const int b = blockIdx.y, t0 = blockIdx.x * PP_TT;
  const int lt = threadIdx.x & 31, grp = threadIdx.x >> 5;       // 8 groups of 32 lanes, lane -> frame
  const int t = t0 + lt;
  const bool live = t < T;
  const float* pb = pred + (size_t)b * PP_C * T + (live ? t : T - 1);
  auto stat = [&](int c) { return c < 6 ? c : (c < 276 ? cidx[c - 6] : n_stats - PP_E + (c - 276)); };
  auto load = [&](int c) {
    float v = pb[(size_t)c * T];
    if (mean) {
      const int s = stat(c);
      v = v * stdv[s] + mean[s];
    }
    return v;
  };
  for (int j = grp; j < PP_J; j += 8) {
    float a[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) a[e] = load(6 * j + e);
    // b1 = a1 / max(|a1|, 1e-12)
    float n1 = sqrtf(__builtin_fmaf(a[2], a[2], __builtin_fmaf(a[1], a[1], a[0] * a[0])));
    n1 = n1 > 1e-12f ? n1 : 1e-12f;
    const float b1x = a[0] / n1, b1y = a[1] / n1, b1z = a[2] / n1;
    const float d = (b1x * a[3] + b1y * a[4]) + b1z * a[5];
    float cx = a[3] - d * b1x, cy = a[4] - d * b1y, cz = a[5] - d * b1z;
    float n2 = sqrtf(__builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, cx * cx)));
    n2 = n2 > 1e-12f ? n2 : 1e-12f;
    const float b2x = cx / n2, b2y = cy / n2, b2z = cz / n2;
    const float b3x = __builtin_fmaf(b1y, b2z, -(b1z * b2y)), b3y = __builtin_fmaf(b1z, b2x, -(b1x * b2z)),
                b3z = __builtin_fmaf(b1x, b2y, -(b1y * b2x));
    // rows b1, b2, b3: m00 = b1x, m11 = b2y, m22 = b3z
    auto sp = [](float x) { return x > 0.f ? sqrtf(x) : 0.f; };
    const float q0 = 0.5f * sp(((1.f + b1x) + b2y) + b3z);
    float qx = 0.5f * sp(((1.f + b1x) - b2y) - b3z);
    float qy = 0.5f * sp(((1.f - b1x) + b2y) - b3z);
    float qz = 0.5f * sp(((1.f - b1x) - b2y) + b3z);
    qx = (b3y - b2z) < 0.f ? -qx : qx;          // _copysign(x, m21 - m12)
    qy = (b1z - b3x) < 0.f ? -qy : qy;          // m02 - m20
    qz = (b2x - b1y) < 0.f ? -qz : qz;          // m10 - m01
    const float nv = sqrtf(__builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx)));
    const float half = atan2f(nv, q0);
    const float ang = 2.f * half;
    const float s = fabsf(ang) < 1e-6f ? 0.5f - (ang * ang) / 48.f : sinf(half) / ang;
    float* row = tile + lt * W;
    row[pp_out_col(3 * j + 0, full)] = qx / s;
    row[pp_out_col(3 * j + 1, full)] = qy / s;
    row[pp_out_col(3 * j + 2, full)] = qz / s;
  }
  // expression coefficients: channels 276..375 -> the last 100 columns
  for (int c = grp; c < PP_E; c += 8) tile[lt * W + (W - PP_E) + c] = load(276 + c);
  if (full) {
    for (int k = grp; k < 27; k += 8) {
      const int col = k < 15 ? 3 + k : (k < 21 ? 21 + (k - 15) : 30 + (k - 21));
      tile[lt * W + col] = lower[k];
    }
  }
  __syncthreads();
  const int rows = T - t0 < PP_TT ? T - t0 : PP_TT;
  if (ld == W) {
    float* ob = out + ((size_t)b * T + t0) * W;
    for (int i = threadIdx.x; i < rows * W; i += 256) ob[i] = tile[i];
  } else {
    for (int i = threadIdx.x; i < rows * W; i += 256) {
      const int r = i / W, c = i - r * W;
      out[((size_t)b * T + t0 + r) * ld + c] = tile[i];
    }
  }
}

int launch_postprocess(const float* pred, int B, int T, const float* mean, const float* stdv, const int* cidx, int n_stats,
                       const float* lower, float* out, int ld, hipStream_t s) {
  PT_CHECK(pred && out && B > 0 && T > 0, "postprocess: null argument or empty batch");
  PT_CHECK((mean == nullptr) == (stdv == nullptr), "postprocess: mean and std come together");
  PT_CHECK(!mean || (cidx && n_stats >= 276 + 54), "postprocess: denormalisation needs c_index (270 entries) and >= 330 statistics");
  PT_CHECK(ld >= (lower ? 265 : 238), "postprocess: output rows of %d floats need ld >= that (got %d)", lower ? 265 : 238, ld);
  PT_CHECK(B <= 65535, "postprocess: B=%d exceeds the grid's y extent", B);
  PT_LAUNCH(postprocess_kernel, dim3((T + PP_TT - 1) / PP_TT, B), dim3(256), 0, s, pred, mean, stdv, cidx, n_stats, lower, out, T, ld);
  return 0;
}

}  // namespace pt
