// Shared declarations for the ProbTalk MI355X (gfx950) hot-path library.
// Internal header: the public C-ABI is include/probtalk_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace pt {

// Activation layout inside the library: channels-last, [B*L, ld] fp32 rows with
// ld = round_up(C, 32) and columns C..ld-1 kept at zero, so every GEMM runs
// without K-tail guards.
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int ld_of(int C) { return round_up(C, 32); }

void set_error(const char* fmt, ...);
const char* last_error();

#define PT_CHECK(cond, ...)                                   \
  do {                                                        \
    if (!(cond)) {                                            \
      ::pt::set_error(__VA_ARGS__);                           \
      return 1;                                               \
    }                                                         \
  } while (0)

#define PT_HIP(expr)                                                             \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) {                                                      \
      ::pt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

// Kernel launch + launch-error check.  hipGetLastError() is per-thread sticky state shared with the
// host framework's own HIP calls (PyTorch leaves benign errors behind, e.g. from device probing), so
// it is cleared first: only this launch's status is reported.
#define PT_LAUNCH(kernel, grid, block, smem, stream, ...)                   \
  do {                                                                      \
    (void)hipGetLastError();                                                \
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);     \
    PT_HIP(hipGetLastError());                                              \
  } while (0)

// ---------------------------------------------------------------------------
// tap-GEMM: C[orow(m), n] = act( sum_t sum_k A[src(m,t), k] * W[t][n][k] + bias[n] (+pe) (+Res) )
// Covers Linear, 1x1 conv, k3/k7 s1 conv, k4 s2 conv and (per output parity) k4 s2 ConvTranspose.
// ---------------------------------------------------------------------------
enum Act { ACT_NONE = 0, ACT_LEAKY = 1, ACT_GATED = 2 };
constexpr int MAX_TAPS = 8;

struct GemmArgs {
  const float* A;   int lda;     // [B*Lin, lda]
  const float* W;                // [slabs][Npad][K]
  const float* bias;             // [N] (gated: [N] in permuted order) or nullptr
  const float* Res; int ldr;     // residual, indexed by OUTPUT row, or nullptr
  float* C;         int ldc;     // output rows of ldc floats
  const double* pe; int pe_period;  // optional fp64 table [pe_period][N]; added in fp64 then rounded (PTransformer.py:679-683)
  int M, N, K, Npad;             // M = B*Lgrid GEMM rows; N real out channels (gated: 2*out); K multiple of 32
  int ntaps;
  int tap_d[MAX_TAPS];           // source frame offset per tap
  int tap_w[MAX_TAPS];           // weight slab per tap
  int Lgrid, Lin, Lout;          // rows per clip: GEMM grid / source / destination
  int in_stride, out_stride, out_off;
  int act; float slope;
  int zero_to;                   // columns [Nout, zero_to) of C are written as zeros
  double flops;                  // algorithmic FLOPs of this launch (profiling only; 0 = 2*M*N*K*taps)
};
int launch_gemm(const GemmArgs& a, hipStream_t s);
constexpr int PROF_CFGS = 8;
void prof_enable(bool on);
int prof_read(double* ms, double* flops, int* launches);

// bf16-operand variant (gemm_bf16.hip): A and W are bf16 bit patterns, accumulation and epilogue fp32,
// output fp32 (C) and/or bf16 (Cb).  K is a multiple of 64.
struct GemmArgsB {
  const unsigned short* A; int lda;
  const unsigned short* W;
  const float* bias;
  const float* Res; int ldr;          // fp32 residual (rows of ldr) ...
  const unsigned short* ResB;         // ... or bf16 residual (same ldr); at most one of the two
  float* C; int ldc;
  unsigned short* Cb; int ldcb;
  const double* pe; int pe_period;
  int M, N, K, Npad;
  int ntaps;
  int tap_d[MAX_TAPS];
  int tap_w[MAX_TAPS];
  int Lgrid, Lin, Lout;
  int in_stride, out_stride, out_off;
  int act; float slope;
  int zero_to;
  double flops;
};
int launch_gemm_bf16(const GemmArgsB& a, hipStream_t s);
// phase-pipelined 256 x 256 LDS-DMA variant for the conv stacks (gemm_bf16_conv.hip); launch_gemm_bf16 dispatches to it
int conv8_prepare();                        // allocates the per-device zero block (model finalize time)
bool conv8_takes(const GemmArgsB& a);
int launch_conv8(const GemmArgsB& a, hipStream_t s);

// Large-tile bf16 Linear (gemm_bf16_big.hip): 256x256 tiles, or 128x512 tiles that own whole d_model=512
// rows and can fuse the LayerNorm that follows the Linear in the decoder layer.
enum Epi { EPI_STD = 0, EPI_LNMOD = 1, EPI_ADDLN = 2 };
struct GemmBigArgs {
  const unsigned short* A; int lda;   // [M, lda] bf16
  const unsigned short* W;            // [Npad, K] bf16
  const float* bias;                  // [N]
  int M, N, K, Npad;
  int epi;
  const unsigned short* Wt;           // optional (row kernel only): W K-tile-major, [K / 32][512][32]; same values as W
  // EPI_STD: C fp32 and/or Cb bf16 = act(A.W^T + bias (+pe fp64) (+Res))
  float* C; int ldc;
  unsigned short* Cb; int ldcb;       // bf16 output (all epilogues)
  const float* Res; int ldr;
  const double* pe; int pe_period;
  int relu, zero_to;
  // EPI_LNMOD: Cb = SiLU(norm(acc+bias) * style[label][0:512] + style[label][512:1024]); style = folded tables
  //            gamma*(1+scale) | beta*(1+scale)+shift per class (gamma/beta unused)
  // EPI_ADDLN: X = LN(X + acc + bias; gamma,beta) in place (fp32), Cb = bf16 copy;
  //            X == nullptr: bf16 residual stream, Cb = LN(Cb + acc + bias) in place (row kernel only)
  const float* gamma; const float* beta;
  const float* style; const int64_t* label; int n_label, rows_per_clip, n_classes;
  float* X;
  double flops;
};
int launch_gemm_big(const GemmBigArgs& a, hipStream_t s);
bool gemm_sm_takes(const GemmBigArgs& a);
bool gemm_sm_wants(const GemmBigArgs& a);
int launch_gemm_sm(const GemmBigArgs& a, hipStream_t s);

// Fused tail of a decoder sub-layer (gemm_bf16_sub.hip): last Linear -> stylization pre-op -> style Linear -> residual +
// post-LN -> optional next Linear, rows kept in registers between the products.
struct SubArgs {
  const unsigned short* A; int lda; int K1;        // sub-layer input rows [M, lda] bf16; K1 = 512 (out-proj) or 2048 (FFN-2)
  const unsigned short* W1; const float* b1;       // [512][K1] bf16
  const float* fold; const int64_t* label; int n_label, rows_per_clip, n_classes;   // folded style tables [classes][1024]
  const unsigned short* W2p; const float* b2;      // style Linear [512][512], K axis permuted (launch_permute_k32)
  const float* gamma; const float* beta;           // post-LN affine
  unsigned short* X;                               // residual stream rows [M][512] bf16, updated in place
  int M;
};
int launch_sublayer(const SubArgs& a, hipStream_t s);
int sublayer_grid_limit();
int launch_permute_k32(const unsigned short* in, unsigned short* out, int N, int K, hipStream_t s);

// Per-device launch state (device.hip).  The library may be used on several GPUs from one process: the dynamic-LDS
// attribute of a kernel and the CU count belong to the CURRENT device, so both are cached per device ordinal.
constexpr int PT_MAX_DEVICES = 64;
struct LdsAttr { int bytes[PT_MAX_DEVICES]; };            // zero-initialised static per kernel: largest size set so far
int ensure_dyn_lds(const void* kernel, int bytes, LdsAttr& st);   // hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per device
int device_cus();                                          // CUs of the current device (256 if the query fails)
#define PT_ENSURE_LDS(kernel, bytes)                                                        \
  do {                                                                                      \
    static ::pt::LdsAttr lds_attr_;                                                         \
    if (::pt::ensure_dyn_lds(reinterpret_cast<const void*>(kernel), (int)(bytes), lds_attr_)) return 1; \
  } while (0)

// Run-time options: the library's only switches (pt_set_option / pt_get_option); nothing reads the environment.
enum Option {
  OPT_RESID_BF16,          // 1 (default): bf16 mode keeps the residual stream in bf16; 0: fp32 stream with a bf16 shadow
  OPT_FUSED_TAIL,          // -1 (default): fused sub-layer tail where whole rounds of its grid pay; 0 never; 1 wherever it applies
  OPT_SMALL_ROWS,          // rows up to which wide Linears (N != 512) take the 4-wave kernel (default 2048)
  OPT_ROW_TILE,            // 0 (default): row kernel picks its tile height; 2 | 4 | 8 | 10 forces 16 x that many rows
  OPT_ATTN_TRAIN_ROWWISE,  // 1: training attention on the row-wise kernels whatever the shape (default 0)
  OPT_CONV8,               // 1 (default): conv stacks of the bf16 mode on the phase-pipelined LDS-DMA kernel where it applies
  OPT_STORE_OVERLAP,       // 1 (default): the 256 x 256 GEMM stores a tile's upper half between the phases of its last K-tile
  OPT_ROW2,                // 1: N = 512 Linears with at least row2_min_rows rows run two 80-row workgroups per CU (gemm_bf16_row2.hip); default 0 (measured slower)
  OPT_ROW2_MIN_ROWS,       // rows from which the two-workgroups-per-CU row kernel is used (default 20480 = 80 rows x 256 CUs)
  OPT_SPLIT_BF16,          // 1 (default): fp32 tap-GEMMs and weight gradients (fp32 mode, PQ-VAE encoder, train step) run on the bf16 MFMA with both
                           // operands split into three bf16 terms, six products (fp32 accuracy, 2.7 x the fp32 MFMA rate); 0: fp32 MFMA
  OPT_SM_ROWS,             // rows up to which a plain bf16 Linear (bias / ReLU epilogue, K % 128 == 0) runs on the small-batch kernel
                           // (gemm_bf16_sm.hip: 64 x 64 tiles, operands of the whole K in flight); 0 = never
  OPT_W_KTILE,             // 1 (default): the row kernel reads the K-tile-major copy of an N = 512 Linear's weights where the model has one; 0: row-major
  OPT_COUNT
};
int option(Option o);

// layout
int launch_ncl_to_cl(const float* in, float* out, int B, int C, int L, int ldo, int col_off, int zero_to, hipStream_t s);
int launch_ncl_to_cl_b(const float* in, unsigned short* out, int B, int C, int L, int ldo, int col_off, int zero_to,
                       hipStream_t s);
int launch_cl_to_ncl(const float* in, int ldi, int col_off, float* out, int B, int C, int L, hipStream_t s);

// pose post-process (postprocess.hip)
int launch_postprocess(const float* pred, int B, int T, const float* mean, const float* stdv, const int* cidx, int n_stats,
                       const float* lower, float* out, int ld, hipStream_t s);

// product quantizer
// prepared = true: pq_search_prepare(D) has run on this device (model finalize), no attribute call here
int launch_pq_search(const float* z_cl, int ldz, const float* codebooks, int rows, int G, int D, int K,
                     int64_t* idx, float* q_cl, int ldq, hipStream_t s, bool prepared = false);
int pq_search_prepare(int D, int K);
int launch_pq_gather(const void* idx, int idx_is_i64, const float* codebooks, int rows, int G, int D, int K,
                     float* q_cl, int ldq, int zero_to, hipStream_t s);

// Sum over the four lanes {l, l^16, l^32, l^48} (the lanes that share an output row of a 16x16 MFMA accumulator), the
// same in all four.  v_permlane16_swap / v_permlane32_swap on a register paired with itself are the xor-16 / xor-32
// butterflies as plain VALU moves: no trip through the LDS crossbar (ds_bpermute, which __shfl_xor compiles to, costs a
// lgkmcnt wait per step), and the same bits (each step adds the same two values).
__device__ __forceinline__ float row_lanes_sum(float s) {
  // the pair's elements go through scalars: __builtin_bit_cast applied to a vector element directly reads element 0
  // for both (clang 20 / ROCm 7.2)
  const unsigned u = __builtin_bit_cast(unsigned, s);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned a0 = a[0], a1 = a[1];
  const float t = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
  const unsigned w = __builtin_bit_cast(unsigned, t);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  const unsigned b0 = b[0], b1 = b[1];
  return __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
}

}  // namespace pt
