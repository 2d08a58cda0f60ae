/* probtalk_hip.h — C ABI of libprobtalk_hip.so, the MI355X (gfx950) implementation of ProbTalk's
 * token hot path (PQ-VAE encode/quantize/decode, predictor MaskGIT transformer, refiner transformer).
 *
 * Conventions (all entry points):
 *   - return 0 on success, non-zero on error; pt_last_error() returns the message (thread local);
 *   - tensor pointers are DEVICE pointers to contiguous fp32 data in the reference's own layout
 *     (NCL for (B,C,T) tensors) unless a parameter says otherwise; int tensors as documented;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*), nothing synchronises;
 *   - no entry point below the "model lifetime" block allocates: scratch comes from the caller's
 *     `ws` buffer of at least pt_*_workspace_bytes() bytes; outputs are caller allocated;
 *   - citations `file:line` name the reference interface (feifeifeiliu/probtalk @ 2024_08_07) that
 *     the entry point replaces.
 */
#ifndef PROBTALK_HIP_H
#define PROBTALK_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_model pt_model;

const char* pt_last_error(void);
int pt_version(void);

/* ---- model lifetime (may allocate) --------------------------------------------------------- */
/* kind / cfg:
 *   "vqvae_sc"     {in_dim, emb_dim, num_codes, num_hiddens, groups, 4}   nets/inpainting/vqvae_1d_sc.py:277-298
 *   "vqvae_legacy" {in_dim, emb_dim, num_codes, num_hiddens, 1, 3}        nets/spg/vqvae_1d.py:147-161
 *   "cond_encoder" {in_dim, num_hiddens}                                  nets/inpainting/vqvae_1d_sc.py:70-84
 *   "predictor"    {mot_dim, num_codes, groups, n_classes, n_layers, audio_dim}  nets/inpainting/predictornet.py:13-59
 *   "refiner"      {in_dim, cond_dim, n_classes, n_layers}                nets/inpainting/RTransformer.py:181-207 */
pt_model* pt_model_create(const char* kind, const int* cfg, int ncfg);
/* Hand over one state_dict entry (HOST pointer, copied): the reference's own key names and layouts,
 * i.e. what `module.state_dict()` / a ckpt-*.pth holds (vq_teacher.py:103-114, predictor.py:128-143).
 * dtype: 0 = float32, 1 = float64 (positional-encoding tables). */
int pt_model_set_tensor(pt_model* m, const char* name, const void* host_data, const int64_t* shape, int ndim, int dtype);
/* Arithmetic: 0 = fp32 (default; parity path, 1e-4), 1 = bf16 MFMA operands with fp32 accumulation (BASELINE configs[2];
 * tolerance 1e-2 as l2-relative error): predictor / refiner transformers (LayerNorm statistics from the fp32
 * accumulators; the residual stream itself is bf16 unless option "resid_bf16" is 0), the motion-context and audio
 * condition encoders, and the PQ-VAE DECODER.  PQ-VAE encode + nearest-code search always run in fp32 (bit-exact
 * indices).  Not available for "vqvae_legacy".  Call before pt_model_finalize. */
int pt_model_set_precision(pt_model* m, int precision);
/* Fold BatchNorm(eval) into the convs, repack, upload.  Replaces nn.Module.load_state_dict + .to(device). */
int pt_model_finalize(pt_model* m);
void pt_model_destroy(pt_model* m);

/* ---- PQ-VAE ------------------------------------------------------------------------------- */
size_t pt_vqvae_workspace_bytes(const pt_model* m, int B, int T);
/* VQVAE_SC.encode (vqvae_1d_sc.py:317-320) + ProductQuantization.forward eval (vqvae_modules.py:470-492):
 * x (B,in_dim,T) -> z (B,emb,T') [optional], quantized (B,emb,T') [optional], indices (B,T',G) int64.
 * Legacy VQVAE.encode (vqvae_1d.py:175-178) with kind "vqvae_legacy" (G = 1). */
int pt_vqvae_encode(pt_model* m, const float* x, int B, int T, float* z, float* quantized, int64_t* indices,
                    void* ws, size_t ws_bytes, void* stream);
/* VQVAE_SC.decode (vqvae_1d_sc.py:322-324): e (B,emb,T') -> out (B,in_dim,8T'). */
int pt_vqvae_decode(pt_model* m, const float* e, int B, int Tq, float* out, void* ws, size_t ws_bytes, void* stream);
/* quantize_all + transpose + decode (predictor.py:439-442; demo.py:152-153): codes (B,T',G) -> out. */
int pt_vqvae_decode_codes(pt_model* m, const void* codes, int codes_are_int64, int B, int Tq, float* out, void* ws,
                          size_t ws_bytes, void* stream);
/* ProductQuantization.forward eval on a caller-supplied latent: z (B,emb,W) -> quantized (B,emb,W), indices (B,W,G). */
int pt_pq_quantize(pt_model* m, const float* z, int B, int W, float* quantized, int64_t* indices, void* ws,
                   size_t ws_bytes, void* stream);
/* pt_pq_quantize for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): no counterpart in the reference.  Same inputs, outputs and results as pt_pq_quantize,
 * but the two channels-last staging images live in a 16 MiB device array that is part of the library image (one copy per
 * GPU, held until the process ends) instead of a caller workspace; the call allocates nothing.  Limit: B*W <= 4096 latent
 * rows and emb_dim <= 512, otherwise it returns non-zero with a message naming the limit.  Every call on one GPU uses that
 * GPU's array: calls that overlap on one GPU (two streams, two threads) race; calls on one stream are safe. */
int pt_pq_quantize_static(pt_model* m, const float* z, int B, int W, float* quantized, int64_t* indices, void* stream);
/* ProductQuantization.quantize_all (vqvae_modules.py:557-564): idx (rows,G) -> out (rows, emb) row-major. */
int pt_pq_quantize_all(pt_model* m, const void* idx, int idx_are_int64, int rows, float* out, void* stream);

/* ---- audio condition encoder -------------------------------------------------------------- */
size_t pt_cond_encoder_workspace_bytes(const pt_model* m, int B, int T);
/* ConditionEncoder.forward (vqvae_1d_sc.py:86-91): aud (B,768,T) -> (B,256,T/8). */
int pt_cond_encoder_forward(pt_model* m, const float* aud, int B, int T, float* out, void* ws, size_t ws_bytes,
                            void* stream);

/* ---- predictor ---------------------------------------------------------------------------- */
size_t pt_predictor_workspace_bytes(const pt_model* m, int B, int T);
/* PredictorNet.forward up to the decoder call (predictornet.py:87-100):
 * masked_motion (B,mot,T), mask (B,1,T), audio (B,256,T') -> condition (B,768,T'). */
int pt_predictor_condition(pt_model* m, const float* masked_motion, const float* mask, const float* audio, int B, int T,
                           float* condition, void* ws, size_t ws_bytes, void* stream);
/* PTransformer.tokens_to_logits (PTransformer.py:657-694): ids (B,T',G) int32 (mask id = num_codes),
 * label (B) int64, condition (B,768,T') -> logits (B,T',G,num_codes). */
int pt_predictor_logits(pt_model* m, const int32_t* ids, const int64_t* label, const float* condition, int B, int Tq,
                        float* logits, void* ws, size_t ws_bytes, void* stream);
/* PTransformer.sample_good (PTransformer.py:590-642): the whole MaskGIT loop with the two random draws
 * injected: noise_q (steps,B,N,num_codes) ~ Exp(1) for the categorical draw, noise_g (steps,B,N) ~ Gumbel(0,1).
 * codes_out (B,T',G) int32.  logits_out (steps,B,N,num_codes) optional (may be NULL). */
int pt_predictor_sample(pt_model* m, const int64_t* label, const float* condition, const float* noise_q,
                        const float* noise_g, int steps, int B, int Tq, int32_t* codes_out, float* logits_out, void* ws,
                        size_t ws_bytes, void* stream);
/* The same loop with both draws generated inside the MaskGIT step kernel (counter-based: splitmix64 of (seed, step,
 * element) -> 24-bit uniform -> Exp(1) / Gumbel(0,1)): no noise tensors in HBM and no host RNG call per pass (the
 * reference draws the Gumbel noise on the CPU and copies it each pass, PTransformer.py:697-698).  Same seed, same codes. */
int pt_predictor_sample_seeded(pt_model* m, const int64_t* label, const float* condition, unsigned long long seed, int steps,
                               int B, int Tq, int32_t* codes_out, float* logits_out, void* ws, size_t ws_bytes, void* stream);
/* pt_predictor_sample_seeded with the seed read from device memory by the kernels (one uint64 at seed_dev): the call
 * bakes no per-call value into its launches, so a caller can capture it into a HIP graph once and replay it with a new
 * seed each time by writing seed_dev first.  No counterpart in the reference (eager PyTorch, host RNG per pass). */
int pt_predictor_sample_seed_dev(pt_model* m, const int64_t* label, const float* condition, const unsigned long long* seed_dev,
                                 int steps, int B, int Tq, int32_t* codes_out, float* logits_out, void* ws, size_t ws_bytes,
                                 void* stream);
/* The noise pt_predictor_sample_seeded uses in pass `step`, written out: noise_q (B,N,V), noise_g (B,N).  Feeding these to
 * pt_predictor_sample reproduces the seeded codes (test hook; also a stand-alone Exp(1) / Gumbel generator). */
int pt_maskgit_noise(unsigned long long seed, int step, int B, int N, int V, float* noise_q, float* noise_g, void* stream);
/* The generator's last stage on caller-chosen 23-bit patterns: u = (bits + 0.5) / 2^23 in [2^-24, 1 - 2^-24], noise_q =
 * -log u, noise_g = -log(-log u), n values each (test hook: the extremes of the range must stay finite; torch's Gumbel
 * likewise clamps its uniform away from 0 and 1, PTransformer.py:697-698). */
int pt_maskgit_noise_from_bits(const uint32_t* bits, int n, float* u, float* noise_q, float* noise_g, void* stream);

/* ---- refiner ------------------------------------------------------------------------------ */
size_t pt_refiner_workspace_bytes(const pt_model* m, int B, int T);
/* RTransformer.forward (RTransformer.py:209-230): x (B,in,T), condition (B,768,T), mask (B,1,T) in {0,1},
 * label (B) or (1) int64 -> out (B,in,T). */
int pt_refiner_forward(pt_model* m, const float* x, const float* condition, const float* mask, const int64_t* label,
                       int n_label, int B, int T, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- glue kernels either side of the path -------------------------------------------------- */
/* F.interpolate(pred, size=T_out, mode='linear') then gt*mask*mask' + pred*(1-mask)
 * (predictor.py:445-446 with premask=1: first term (gt*mask)*mask; refiner.py:528-529,553 with premask=0: gt*mask).
 * pred (B,C,T_in), gt (B,C,T_out), mask (B,1,T_out) -> out (B,C,T_out). gt/mask may be NULL (pure resize). */
int pt_interp_blend(const float* pred, int B, int C, int T_in, int T_out, const float* gt, const float* mask,
                    int premask, float* out, void* stream);

/* ---- op-level entry points (unit tests, reuse) -------------------------------------------- */
/* Conv1d / ConvTranspose1d on channels-last rows through the tap-GEMM: A [B*Lin, lda], W packed
 * [k][Npad][Kpad], out [B*Lout, ldc].  kind: 0=k1 1=k3 2=k7 3=k4s2 4=convT k4s2. act: 0 none 1 leaky(slope) 2 gated. */
int pt_conv1d_cl(const float* A, int lda, const float* Wpacked, const float* bias, int N, int Npad, int K, int kind,
                 int B, int Lin, float* C, int ldc, int act, float slope, const float* Res, int ldr, void* stream);
/* The same conv on bf16 operands (bf16 mode of the conv stacks): A [B*Lin, lda] bf16 channels-last, Wpacked
 * [taps][Npad][K] bf16 (K % 64 == 0), fp32 accumulation; output fp32 rows (C) and/or bf16 rows (Cb); optional bf16
 * residual indexed by output row.  Large shapes run on the phase-pipelined LDS-DMA tap-GEMM (option "conv8").
 * The FIRST call on a device allocates a 256-byte zero block (kept until the process ends) unless a bf16 model has been
 * finalized on that device before: do not make that first call under stream capture. */
int pt_conv1d_cl_bf16(const void* A, int lda, const void* Wpacked, const float* bias, int N, int Npad, int K, int kind, int B,
                      int Lin, float* C, int ldc, void* Cb, int ldcb, int act, float slope, const void* ResB, int ldr,
                      void* stream);
int pt_ncl_to_cl(const float* in, float* out, int B, int C, int L, int ldo, void* stream);
int pt_cl_to_ncl(const float* in, int ldi, float* out, int B, int C, int L, void* stream);
/* nearest-code search on channels-last rows: z [rows, ldz], codebooks [G][K][D] -> idx (rows,G), q [rows, ldq]. */
int pt_pq_search_cl(const float* z, int ldz, const float* codebooks, int rows, int G, int D, int K, int64_t* idx,
                    float* q, int ldq, void* stream);

/* bf16-operand Linear: C[M,N] (fp32) and/or Cb[M,N] (bf16) = act(A[M,K] . W[N,K]^T + bias); A, W bf16 bit patterns,
 * W rows padded to Npad (multiple of 128), K a multiple of 64 (nn.Linear / 1x1 conv of the bf16 mode).
 * N >= 256 runs on the large-tile LDS-DMA kernel, narrower outputs on the 4-wave tap-GEMM. */
int pt_gemm_bf16(const void* A, int lda, const void* W, const float* bias, int M, int N, int K, int Npad, float* C,
                 int ldc, void* Cb, int ldcb, int relu, void* stream);
/* Linear to d_model = 512 with the stylization pre-op fused (DeepSupervisionLayer sub-layer output -> StylizationBlock,
 * PTransformer.py:134-139): out_bf16[M,512] = SiLU(LN(A.W^T + bias; gamma, beta) * (1 + scale[c]) + shift[c]),
 * c = label[row / rows_per_clip] (label[0] when n_label == 1).  The LayerNorm affine comes folded into the per-class
 * table style_folded [n_classes][1024] = gamma*(1+scale[c]) | beta*(1+scale[c]) + shift[c]. */
int pt_gemm_bf16_lnmod(const void* A, int lda, const void* W, const float* bias, int M, int K, int Npad,
                       const float* style_folded, const int64_t* label, int n_label, int rows_per_clip, int n_classes,
                       void* out_bf16, void* stream);
/* Linear to d_model = 512 with the post-LN residual fused (PTransformer.py:329-331):
 * x[M,512] = LN(x + A.W^T + bias; gamma, beta) in place (fp32), x_bf16 = its bf16 copy.
 * x == NULL: bf16 residual stream, x_bf16 = LN(x_bf16 + A.W^T + bias) in place (one bf16 rounding per sub-layer on the
 * residual path; 1 KB read + 1 KB written per row instead of 2 KB + 3 KB). */
int pt_gemm_bf16_addln(const void* A, int lda, const void* W, const float* bias, int M, int K, int Npad,
                       const float* gamma, const float* beta, float* x, void* x_bf16, void* stream);
/* The whole tail of a decoder sub-layer in one launch, rows kept in registers between the products (DeepSupervisionLayer,
 * PTransformer.py:329-331 around StylizationBlock_V3 'sb', :134-140):
 *   t = A.W1^T + b1 (A [M, lda] bf16, W1 [512][K1] bf16, K1 = 512 attention out-proj / 2048 FFN-2)
 *   h = SiLU(LN(t) folded with the per-class style table, as pt_gemm_bf16_lnmod)
 *   x_bf16[M,512] = LN(x_bf16 + h.W2^T + b2; gamma, beta) in place (bf16 residual stream, as pt_gemm_bf16_addln with x == NULL)
 * W2p is [512][512] bf16 with the K axis permuted by pt_permute_k32 (the accumulator layout of one product is the operand
 * layout of the next up to that permutation).  n_classes <= 4.  The model path uses it where whole rounds of its
 * persistent grid beat the two launches above. */
int pt_gemm_bf16_sublayer(const void* A, int lda, int K1, const void* W1, const float* b1, const float* style_folded,
                          const int64_t* label, int n_label, int rows_per_clip, int n_classes, const void* W2p, const float* b2,
                          const float* gamma, const float* beta, void* x_bf16, int M, void* stream);
/* out[n][32 kb + 8 q + j] = W[n][32 kb + (j < 4 ? 4 q + j : 16 + 4 q + (j - 4))] for a [N][K] bf16 matrix, K % 32 == 0. */
int pt_permute_k32(const void* W, void* out, int N, int K, void* stream);
/* softmax(Q K^T / 8) V per (clip, head), head dim 64 (nn.MultiheadAttention core, PTransformer.py:340-353):
 * Q rows b*Nq+i at Q + row*ldq + head*64; K/V rows b*Nk+j likewise; O like Q. */
int pt_attention_cl(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, int B,
                    int Nq, int Nk, int H, void* stream);
/* the same core in the bf16 mode: Q/K/V/O are bf16 bit patterns (leading dims in elements, multiples of 8, 16-byte aligned
 * bases), bf16 MFMA with fp32 softmax statistics and accumulation; K and V of a (clip, head) stay resident in LDS up to
 * 544 keys and stream through it in 256-key chunks beyond (any Nk). */
int pt_attention_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int B,
                      int Nq, int Nk, int H, void* stream);
/* measurement aid (bench.py roofline leg): when enabled every tap-GEMM launch is bracketed by HIP events
 * on its stream; pt_prof_read sums elapsed ms / algorithmic FLOPs / launches per kernel config
 * {0-2: fp32 tap-GEMM 128x128 / 64x128 / 64x64 tiles, 3-5: 4-wave bf16 tap-GEMM 128x128 / 64x128 / 64x64 and the conv
 * stacks' 256x256 LDS-DMA kernel, 6: bf16 256x256 Linears (gemm8), 7: bf16 N = 512 row-complete Linears with their
 * LayerNorm epilogues (row kernels, fused sub-layer tail)} (arrays of 8);
 * call it after synchronising. */
int pt_prof_enable(int on);
int pt_prof_read(double* ms8, double* flops8, int* launches8);
/* code gather on raw buffers: idx (rows,G) int32/int64, codebooks [G][K][D] -> out [rows, ldq] (quantize_all). */
int pt_pq_gather_cl(const void* idx, int idx_are_int64, const float* codebooks, int rows, int G, int D, int K,
                    float* out, int ldq, void* stream);

/* ---- training-mode operators (train.hip): predictor train step of predictor.py:168-280 ---------------------------
 * All tensors are channels-last fp32 rows [B*L, ld] as in the op-level block above.  The forward GEMMs and the
 * input-gradient GEMMs of the train step are pt_conv1d_cl itself (input gradient = the same tap-GEMM on dY with
 * transposed weights, see train.hip); what is added here is everything else the backward pass and the optimizer need.
 * `kind` as in pt_conv1d_cl. */
/* token embedding + fp64 positional encodings (PTransformer.py:447-474): x[row] = float(double(tok[id]) + (pe[t+1] + ppe[g+1])) */
int pt_embed_pe(const int32_t* ids, const float* tok, const double* pe, const double* ppe, float* x, int rows, int N, int G,
                int vocab, void* stream);
/* weight gradient of a conv/linear (the `.grad` autograd gives nn.Conv1d / nn.Linear weights): dW [taps][N][Cin] (pre-zeroed) +=
 * sum over rows of dY[row][n] * X[tap-shifted row][c] */
int pt_train_wgrad(const float* dY, int lddy, const float* X, int ldx, float* dW, int N, int Cin, int kind, int B, int Lin,
                   void* stream);
/* the same plus the bias gradient db[N] (pre-zeroed) = column sums of dY (nn.Conv1d / nn.Linear .bias.grad): accumulated
 * inside the weight-gradient kernel from the dY tiles it stages anyway (option split_bf16 = 1), else pt_train_colsum's
 * launch behind it. */
int pt_train_wgrad_bias(const float* dY, int lddy, const float* X, int ldx, float* dW, float* db, int N, int Cin, int kind, int B,
                        int Lin, void* stream);
/* out[c] += sum_rows a[row][c] (* b[row][c] if b): bias gradients */
int pt_train_colsum(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* out, void* stream);
/* nn.LayerNorm(512) on x (+ res): y, stats[rows][2] = mean, rstd; backward gives dx (w.r.t. x + res), dgamma, dbeta (+=) */
int pt_train_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y, float* stats, int rows,
                    void* stream);
int pt_train_ln_bwd(const float* dy, const float* x, const float* res, const float* stats, const float* gamma, float* dx,
                    float* dgamma, float* dbeta, int rows, void* stream);
/* StylizationBlock_V3 'sb' modulation (PTransformer.py:134-139): h = SiLU(y*(1+scale[c]) + shift[c]), table [n_classes][1024] */
int pt_train_modsilu_fwd(const float* y, const float* table, const int64_t* label, int n_label, int rows_per_clip, int n_classes,
                         float* h, int rows, void* stream);
int pt_train_modsilu_bwd(const float* dh, const float* y, const float* table, const int64_t* label, int n_label,
                         int rows_per_clip, int n_classes, float* dy, float* dtable, int rows, void* stream);
/* nn.MultiheadAttention core in train mode (dropout p on the attention weights, mask = hash(seed, b, h, i, j)) and its
 * backward (probabilities recomputed; lse / delta are [B*H*Nq] scratch outputs) */
int pt_train_attention_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, int B,
                           int Nq, int Nk, int H, float pdrop, uint64_t seed, void* stream);
int pt_train_attention_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* dO, int ldo,
                           float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, float* lse, float* delta, int B,
                           int Nq, int Nk, int H, float pdrop, uint64_t seed, void* stream);
/* F.cross_entropy(logits [rows,V], target) mean (predictor.py:261): *loss (pre-zeroed) and dlogits */
int pt_train_cross_entropy(const float* logits, const int64_t* target, int V, int rows, float* loss, float* dlogits,
                           void* stream);
/* nn.BatchNorm1d in train mode + residual + (leaky) ReLU of ConvNormRelu (vqvae_modules.py:193-200): batch statistics over
 * the rows, running statistics updated in place (may be NULL); slope < 0 = no activation.  acc_ws: 2*cols doubles. */
int pt_train_bn_fwd(const float* x, int ldx, int rows, int cols, const float* gamma, const float* beta, const float* res, int ldr,
                    float slope, float* y, int ldy, float* mean_rstd, double* acc_ws, float* run_mean, float* run_var,
                    float momentum, void* stream);
/* backward: dz = gradient before the activation (also the residual branch's gradient), dx, sums [2][cols] = dbeta | dgamma */
int pt_train_bn_bwd(const float* dy, int lddy, const float* y, int ldy, float slope, const float* x, int ldx,
                    const float* mean_rstd, const float* gamma, float* dz, int lddz, float* dx, int lddx, float* sums, int rows,
                    int cols, void* stream);
int pt_train_act_bwd(const float* dy, int lddy, const float* y, int ldy, float slope, float* dz, int lddz, int rows, int cols,
                     void* stream);
/* GatedActivation (gated_pixelcnn_1d.py:30-36): out = tanh(a) * sigmoid(b), hv rows = [a | b] */
int pt_train_gate_fwd(const float* hv, int rows, int C, float* out, void* stream);
int pt_train_gate_bwd(const float* dout, const float* hv, int rows, int C, float* dhv, void* stream);
/* nn.Embedding gradient: dE[ids[row]] += dx[row] (512 wide) */
int pt_train_embed_bwd(const float* dx, const int32_t* ids, int rows, int vocab, float* dE, void* stream);
/* nn.Dropout: y = x * mask / (1-p), mask = hash(seed, element) (call again on dy with the same seed for the backward) */
int pt_train_dropout(const float* x, float* y, size_t n, float p, uint64_t seed, void* stream);
/* torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (predictor.py:112-118,270-278): *out += sum g^2; coef = min(1,
 * max_norm / (sqrt(sumsq) + 1e-6)); AdamW update with the gradient scaled by *gscale (device scalar, may be NULL) */
int pt_train_sumsq(const float* g, size_t n, float* out, void* stream);
int pt_train_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream);
int pt_train_adamw(float* p, const float* g, float* m, float* v, size_t n, const float* gscale, float lr, float beta1,
                   float beta2, float eps, float wd, int step, void* stream);
int pt_train_add(const float* a, const float* b, float* out, size_t n, void* stream);
/* ProductQuantization.forward train branch (vqvae_modules.py:495-534): EMA cluster sizes / code sums from the assigned
 * latent rows, Laplace smoothing, dead-code re-seeding from the first K latent rows, new codebooks.  z rows [rows, ldz]
 * (G*D used columns), idx int64 [rows, G]; counts [G,K] and dw [G,K,D] are scratch; cs_hidden [G,K], dw_hidden [G,K,D],
 * emb [G,K,D] are the module's buffers (ema_cluster_size.i.hidden, ema_dw.i.hidden, embeddings_i) stacked over i;
 * counter = the EMA modules' call count including this call.
 * rows < K: `_tile` (vqvae_modules.py:571-580) repeats the batch and adds N(0,1) * 0.01 / sqrt(D): seed row k = chunk
 * k mod rows + noise.  tile_noise [G][K][D] (device) injects the N(0,1) draws (row k of group g = row k of the g-th
 * torch.randn_like of the reference: the parity mode); NULL = generated in the kernel from tile_seed. */
int pt_train_pq_ema(const float* z, int ldz, const int64_t* idx, int rows, int G, int D, int K, float* counts, float* dw,
                    float* cs_hidden, float* dw_hidden, float* emb, float decay, float eps, int counter, const float* tile_noise,
                    uint64_t tile_seed, void* stream);
/* The same with share_code=True (vqvae_modules.py:496-512): the chunks of all G groups update ONE codebook.  counts [K] and
 * dw [K,D] are scratch; cs_hidden [K], dw_hidden [K,D], emb [K,D] are ema_cluster_size.hidden, ema_dw.hidden and
 * embeddings_0.  Dead codes are re-seeded from the first K chunks in group-major order (all rows of group 0, then group
 * 1, ...: line 498's flat_x); the EMA state is NOT reset (the share_code branch has no lines 529-532).  rows*G < K: tiled
 * with noise as above, tile_noise [K][D]. */
int pt_train_pq_ema_shared(const float* z, int ldz, const int64_t* idx, int rows, int G, int D, int K, float* counts, float* dw,
                           float* cs_hidden, float* dw_hidden, float* emb, float decay, float eps, int counter, const float* tile_noise,
                           uint64_t tile_seed, void* stream);
/* VectorQuantizerEMA.forward train branch (vqvae_modules.py:323-334), the legacy single codebook of s2g_body_vq: the same
 * EMA update for ONE codebook [K, D] without dead-code re-seeding; any number of latent rows; K may exceed 1024. */
int pt_train_vq_ema(const float* z, int ldz, const int64_t* idx, int rows, int D, int K, float* counts, float* dw,
                    float* cs_hidden, float* dw_hidden, float* emb, float decay, float eps, int counter, void* stream);
/* vq_teacher.TrainWrapper.get_loss (vq_teacher.py:184-232) on channels-last rows [B*T, ld]: smooth-L1(beta) mean per
 * column range bounds[p]..bounds[p+1] (host arrays, nparts <= 4), velocity L1 mean over the first vel_cols columns;
 * sums (device, nparts+2 floats) = part means | velocity mean | sum_p weights[p]*part_p + velocity; dpred = gradient of
 * that total. */
int pt_train_vq_losses(const float* pred, int ldp, const float* gt, int ldg, int B, int T, int nparts, const int* bounds,
                       const float* weights, int vel_cols, float beta, float* sums, float* dpred, int lddp, void* stream);
/* refiner.TrainWrapper.get_loss (refiner.py:313-391): per part sum |pred - gt| * nmask[row] / (t_nmask * cols) with
 * nmask = 1 - mask per frame row and t_nmask = sum(nmask); velocity L1 mean (unmasked) over the first vel_cols columns;
 * sums / dpred as in pt_train_vq_losses. */
int pt_train_masked_l1_losses(const float* pred, int ldp, const float* gt, int ldg, const float* nmask, float t_nmask, int B, int T,
                              int nparts, const int* bounds, const float* weights, int vel_cols, float* sums, float* dpred,
                              int lddp, void* stream);
/* commitment loss (vqvae_modules.py:537-541): *out += sum (a-b)^2 ; dz = dq + (*g) * coef * (z - q) (dq, g may be NULL) */
int pt_train_sqdiff(const float* a, const float* b, size_t n, float* out, void* stream);
int pt_train_commit_bwd(const float* dq, const float* z, const float* q, const float* g, float coef, size_t n, float* dz,
                        void* stream);
/* Pose post-process of a generated batch, one kernel (scripts/demo.py:183-205): pred (B,376,T) network output ->
 * out (B*T rows, ld_out floats): optional denormalisation x*std + mean with the statistics gathered per channel as
 * nets/utils.py:16-43 does for shape 'fbhe' (mean / std: n_stats floats, the two halves of data_utils/norm_stats.npy
 * (2,1,1,430); c_index: the 270 int32 of data_utils/lower_body.py:54-63; NULL mean = normalisation off), 6-D ->
 * rotation matrix -> axis-angle for the 46 joints (data_utils/rotation_conversion.py:512-533,97-119,481-509), the 100
 * expression coefficients appended, and part2full's 27 fixed lower-body scalars inserted (data_utils/lower_body.py:66-83;
 * lower_pose NULL = skip, rows of 238 instead of 265). */
int pt_postprocess_poses(const float* pred, int B, int T, const float* mean, const float* std_, const int* c_index,
                         int n_stats, const float* lower_pose, float* out, int ld_out, void* stream);
/* The library's only run-time switches; it reads no environment variables.  Set them before the calls they affect, from
 * one thread (they are plain process-wide ints).  No counterpart in the reference: they select between kernels that
 * compute the same thing (A/B measurements and tests), except resid_bf16, which moves the bf16 mode's rounding points.
 *   "resid_bf16"         1 (default) bf16 mode keeps the residual stream in bf16; 0 = fp32 stream with a bf16 shadow
 *   "fused_tail"         -1 (default) one-launch sub-layer tail (pt_gemm_bf16_sublayer) where whole rounds of its grid
 *                        pay; 0 never; 1 wherever it applies
 *   "small_rows"         rows up to which wide Linears (N != 512) take the 4-wave kernel (default 2048)
 *   "row_tile"           0 (default) the N = 512 row kernel picks its tile height; 2 | 4 | 8 | 10 forces 16 x that many rows
 *   "attn_train_rowwise" 1 = training attention on the row-wise kernels whatever the shape (default 0)
 *   "conv8"              1 (default) conv stacks of the bf16 mode on the phase-pipelined LDS-DMA tap-GEMM where it applies
 *   "store_overlap"      1 (default) the 256 x 256 GEMM stores a bf16 tile's upper half between the phases of its last
 *                        K-tile; 0 = after the loop (same bits either way)
 *   "row2"               1 = N = 512 Linears with at least row2_min_rows rows run as 80-row tiles, two workgroups per CU
 *                        (gemm_bf16_row2.hip: LayerNorm epilogues beside the other workgroup's MFMAs); default 0 = the
 *                        one-workgroup-per-CU row kernel everywhere (bit-identical results; the two-workgroup form
 *                        measured 8-30 % slower per launch)
 *   "row2_min_rows"      rows from which row2 applies (default 20480 = 80 rows x 256 CUs)
 *   "split_bf16"         1 (default) the fp32 tap-GEMM (fp32 mode, PQ-VAE encoder, training forward / input gradients)
 *                        and the weight gradient (pt_train_wgrad) run on the bf16 matrix cores with each fp32 operand
 *                        split into three bf16 terms and six products per pair (fp32 accuracy at 2.7 x the fp32 MFMA
 *                        rate); 0 = the fp32 MFMA kernels
 *   "sm_rows"            4800 (default) rows up to which a plain bf16 Linear (bias / ReLU, K % 128 == 0) runs on the
 *                        small-batch kernel (64 x 64 tiles, the operands of the whole K requested up front); 0 = never
 *   "w_ktile"            1 (default) the N = 512 row kernel streams the K-tile-major copy of a model's weights
 *                        ([K / 32][512][32]: whole cache lines per DMA instruction); 0 = the row-major weights
 * Unknown names and out-of-range values are errors (non-zero return, pt_last_error). */
int pt_set_option(const char* name, int value);
int pt_get_option(const char* name, int* value);

#ifdef __cplusplus
}
#endif
#endif
