// extern "C" surface of libprobtalk_hip.so (declarations: include/probtalk_hip.h).
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/probtalk_hip.h"
#include "model.h"
#include "transformer.h"

using namespace pt;

struct pt_model {
  Model m;
  void* scratch = nullptr;   // vqvae kinds: this device's copy of the static scratch array, resolved at pt_model_create
};

namespace pt {
int launch_wgrad(const float* dY, int lddy, const float* X, int ldx, float* dW, float* db, int N, int Cin, int kind, int B, int Lin, hipStream_t s);
int launch_colsum(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* out, hipStream_t s);
int launch_ln_fwd(const float* x, const float* res, const float* g, const float* b, float* y, float* stats, int rows, hipStream_t s);
int launch_ln_bwd(const float* dy, const float* x, const float* res, const float* stats, const float* g, float* dx, float* dg, float* db, int rows, hipStream_t s);
int launch_modsilu_fwd(const float* y, const float* table, const int64_t* label, int n_label, int rows_per_clip, int n_classes, float* h, int rows, hipStream_t s);
int launch_modsilu_bwd(const float* dh, const float* y, const float* table, const int64_t* label, int n_label, int rows_per_clip, int n_classes, float* dy, float* dtable, int rows, hipStream_t s);
int launch_attention_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* dO, int ldo, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, float* lse, float* delta, int B, int Nq, int Nk, int H, float pdrop, unsigned long long seed, hipStream_t s);
int launch_attention_train_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, int B, int Nq, int Nk, int H, float pdrop, unsigned long long seed, hipStream_t s);
int launch_ce(const float* logits, const int64_t* target, int V, int rows, float* loss, float* dlogits, hipStream_t s);
int launch_bn_train_fwd(const float* x, int ldx, int rows, int cols, const float* g, const float* b, const float* res, int ldr, float slope, float* y, int ldy, float* mean_rstd, double* acc_ws, float* run_mean, float* run_var, float momentum, hipStream_t s);
int launch_bn_train_bwd(const float* dy, int lddy, const float* y, int ldy, float slope, const float* x, int ldx, const float* mean_rstd, const float* g, float* dz, int lddz, float* dx, int lddx, float* sums, int rows, int cols, hipStream_t s);
int launch_act_bwd(const float* dy, int lddy, const float* y, int ldy, float slope, float* dz, int lddz, int rows, int cols, hipStream_t s);
int launch_gate_fwd(const float* hv, int rows, int C, float* out, hipStream_t s);
int launch_gate_bwd(const float* dout, const float* hv, int rows, int C, float* dhv, hipStream_t s);
int launch_embed_bwd(const float* dx, const int* ids, int rows, int vocab, float* dE, hipStream_t s);
int launch_dropout(const float* x, float* y, size_t n, float p, unsigned long long seed, hipStream_t s);
int launch_sumsq(const float* g, size_t n, float* out, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, const float* gscale, float lr, float beta1, float beta2, float eps, float wd, int step, hipStream_t s);
int launch_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, hipStream_t s);
int launch_add(const float* a, const float* b, float* out, size_t n, hipStream_t s);
int launch_attention_bf16(const unsigned short* Q, int ldq, const unsigned short* K, int ldk, const unsigned short* V, int ldv, unsigned short* O, int ldo, int B, int Nq, int Nk, int H, hipStream_t s);
int launch_maskgit_noise(unsigned long long seed, int step, int B, int N, int V, float* q, float* g, hipStream_t s);
int launch_maskgit_bits(const unsigned* bits, int n, float* u, float* q, float* g, hipStream_t s);
int launch_pq_ema(const float* z, int ldz, const int64_t* idx, int rows, int G, int D, int K, float* counts, float* dw, float* cs_hidden, float* dw_hidden, float* emb, float decay, float eps, int counter, hipStream_t s, int reseed, const float* tile_noise, unsigned long long tile_seed);
int launch_vq_losses(const float* pred, int ldp, const float* gt, int ldg, int B, int T, int nparts, const int* bounds, const float* weights, int vel_cols, float beta, float* sums, float* dpred, int lddp, hipStream_t s, const float* nmask, float t_nmask, int l1);
int launch_sqdiff(const float* a, const float* b, size_t n, float* out, hipStream_t s);
int launch_commit_bwd(const float* dq, const float* z, const float* q, const float* g, float coef, size_t n, float* dz, hipStream_t s);
int launch_embed_pe(const int* ids, const float* tok, const double* pe, const double* ppe, float* x, unsigned short* xb, int rows, int N, int G, int vocab, hipStream_t s);
}  // namespace pt

// ---- pt_pq_quantize for callers that cannot allocate: its two channels-last staging images (latent rows in, code rows
// out) live in one file-scope device array instead of a caller workspace.  PT_PQ_STATIC_MAX_ROWS latent rows of up to
// PT_PQ_STATIC_MAX_LD floats each, twice: 16 MiB per GPU, held until the process ends.
#define PT_PQ_STATIC_MAX_ROWS 4096
#define PT_PQ_STATIC_MAX_LD 512
static __device__ __align__(256) unsigned char scratch[2ull * PT_PQ_STATIC_MAX_ROWS * PT_PQ_STATIC_MAX_LD * sizeof(float)];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};   // per device: the address of THAT device's copy of `scratch`

// Device address of the current device's `scratch`, resolved once per device (the array's name is only a host-side
// stand-in for the symbol; every GPU holds its own copy at its own address).
int scratch_for_current_device(void** out) {
  int dev = -1;
  PT_HIP(hipGetDevice(&dev));
  PT_CHECK(dev >= 0 && dev < kMaxDevices, "pt_pq_quantize_static: device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    PT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    PT_CHECK(p != nullptr, "pt_pq_quantize_static: hipGetSymbolAddress returned a null address on device %d", dev);
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}

// ---- run-time options (pt_common.h: Option).  Plain ints: set them before the calls they affect, from one thread.
// One table: name, default, allowed range (row_tile additionally takes only the built tile heights).
struct OptSpec { const char* name; int def, lo, hi; };
constexpr OptSpec g_opt_spec[pt::OPT_COUNT] = {
    {"resid_bf16", 1, 0, 1},          {"fused_tail", -1, -1, 1}, {"small_rows", 2048, 0, 1 << 30},
    {"row_tile", 0, 0, 10},           {"attn_train_rowwise", 0, 0, 1}, {"conv8", 1, 0, 1},
    {"store_overlap", 1, 0, 1},       {"row2", 0, 0, 1},         {"row2_min_rows", 20480, 0, 1 << 30},
    {"split_bf16", 1, 0, 1},          {"sm_rows", 4800, 0, 1 << 30},
    {"w_ktile", 1, 0, 1},
};
int g_opt[pt::OPT_COUNT] = {g_opt_spec[0].def, g_opt_spec[1].def, g_opt_spec[2].def, g_opt_spec[3].def, g_opt_spec[4].def,
                            g_opt_spec[5].def, g_opt_spec[6].def, g_opt_spec[7].def, g_opt_spec[8].def, g_opt_spec[9].def,
                            g_opt_spec[10].def, g_opt_spec[11].def};
static_assert(pt::OPT_COUNT == 12, "option table out of step with pt_common.h");
int find_option(const char* name) {
  if (name)
    for (int i = 0; i < pt::OPT_COUNT; ++i)
      if (strcmp(name, g_opt_spec[i].name) == 0) return i;
  return -1;
}
std::string option_names() {
  std::string all;
  for (int i = 0; i < pt::OPT_COUNT; ++i) all += std::string(i ? ", " : "") + g_opt_spec[i].name;
  return all;
}
}  // namespace

namespace pt {
int option(Option o) { return g_opt[o]; }
}  // namespace pt

extern "C" {

const char* pt_last_error(void) { return last_error(); }
int pt_version(void) { return 100; }

int pt_set_option(const char* name, int value) {
  const int i = find_option(name);
  PT_CHECK(i >= 0, "pt_set_option: unknown option '%s' (options: %s)", name ? name : "(null)", option_names().c_str());
  PT_CHECK(value >= g_opt_spec[i].lo && value <= g_opt_spec[i].hi, "pt_set_option: %s = %d is outside [%d, %d]", name, value,
           g_opt_spec[i].lo, g_opt_spec[i].hi);
  PT_CHECK(i != OPT_ROW_TILE || value == 0 || value == 2 || value == 4 || value == 8 || value == 10, "pt_set_option: row_tile = 0, 2, 4, 8 or 10");
  g_opt[i] = value;
  return 0;
}
int pt_get_option(const char* name, int* value) {
  const int i = find_option(name);
  PT_CHECK(i >= 0 && value, "pt_get_option: unknown option '%s'", name ? name : "(null)");
  *value = g_opt[i];
  return 0;
}

pt_model* pt_model_create(const char* kind, const int* cfg, int ncfg) {
  static const char* kinds[] = {"vqvae_sc", "vqvae_legacy", "cond_encoder", "predictor", "refiner"};
  bool ok = false;
  for (const char* k : kinds) ok = ok || (kind && strcmp(kind, k) == 0);
  if (!ok) {
    set_error("pt_model_create: unknown kind '%s'", kind ? kind : "(null)");
    return nullptr;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("pt_model_create: no HIP device available");
    return nullptr;
  }
  void* scratch_ptr = nullptr;
  if (strncmp(kind, "vqvae", 5) == 0) {   // set-up time: resolve this device's static scratch (pt_pq_quantize_static)
    if (scratch_for_current_device(&scratch_ptr)) return nullptr;
  }
  pt_model* h = new pt_model();
  h->scratch = scratch_ptr;
  h->m.kind = kind;
  h->m.cfg.assign(cfg, cfg + ncfg);
  return h;
}

int pt_model_set_tensor(pt_model* h, const char* name, const void* host_data, const int64_t* shape, int ndim,
                        int dtype) {
  PT_CHECK(h && name && host_data, "pt_model_set_tensor: null argument");
  PT_CHECK(!h->m.finalized, "pt_model_set_tensor: model already finalized");
  PT_CHECK(dtype == 0 || dtype == 1, "pt_model_set_tensor(%s): dtype %d unsupported (0=f32, 1=f64)", name, dtype);
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.dtype = dtype;
  const int64_t n = t.numel();
  PT_CHECK(n >= 0, "pt_model_set_tensor(%s): negative size", name);
  if (dtype == 0) t.f.assign((const float*)host_data, (const float*)host_data + n);
  else t.d.assign((const double*)host_data, (const double*)host_data + n);
  h->m.sd[name] = std::move(t);
  return 0;
}

int pt_model_set_precision(pt_model* h, int precision) {
  PT_CHECK(h, "pt_model_set_precision: null model");
  PT_CHECK(!h->m.finalized, "pt_model_set_precision: model already finalized");
  PT_CHECK(precision == 0 || precision == 1, "pt_model_set_precision: %d unsupported (0 = fp32, 1 = bf16)", precision);
  PT_CHECK(precision == 0 || h->m.kind != "vqvae_legacy", "pt_model_set_precision: the legacy VQ-VAE has no bf16 mode");
  h->m.precision = precision;
  return 0;
}

int pt_model_finalize(pt_model* h) {
  PT_CHECK(h, "pt_model_finalize: null model");
  if (finalize_model(h->m)) return 1;
  // set-up time, current device: the code search's dynamic-LDS attribute, so that pt_pq_quantize_static itself makes no
  // runtime query (it may run under stream capture)
  if (h->scratch && pq_search_prepare(h->m.vq.emb_dim / h->m.vq.groups, h->m.vq.num_codes)) return 1;
  return 0;
}

void pt_model_destroy(pt_model* h) { delete h; }

// ---- PQ-VAE
size_t pt_vqvae_workspace_bytes(const pt_model* h, int B, int T) {
  if (!h || !h->m.finalized) return 0;
  return vqvae_workspace_bytes(h->m, B, T);
}
int pt_vqvae_encode(pt_model* h, const float* x, int B, int T, float* z, float* quantized, int64_t* indices, void* ws,
                    size_t ws_bytes, void* stream) {
  PT_CHECK(h && x && indices, "pt_vqvae_encode: null argument");
  return vqvae_encode(h->m, x, B, T, z, quantized, indices, ws, ws_bytes, (hipStream_t)stream);
}
int pt_vqvae_decode(pt_model* h, const float* e, int B, int Tq, float* out, void* ws, size_t ws_bytes, void* stream) {
  PT_CHECK(h && e && out, "pt_vqvae_decode: null argument");
  return vqvae_decode(h->m, e, nullptr, 0, B, Tq, out, ws, ws_bytes, (hipStream_t)stream);
}
int pt_vqvae_decode_codes(pt_model* h, const void* codes, int codes_are_int64, int B, int Tq, float* out, void* ws,
                          size_t ws_bytes, void* stream) {
  PT_CHECK(h && codes && out, "pt_vqvae_decode_codes: null argument");
  return vqvae_decode(h->m, nullptr, codes, codes_are_int64, B, Tq, out, ws, ws_bytes, (hipStream_t)stream);
}
int pt_pq_quantize(pt_model* h, const float* z, int B, int W, float* quantized, int64_t* indices, void* ws,
                   size_t ws_bytes, void* stream) {
  PT_CHECK(h && z && indices, "pt_pq_quantize: null argument");
  PT_CHECK(h->m.finalized && (h->m.kind == "vqvae_sc" || h->m.kind == "vqvae_legacy"), "pt_pq_quantize: wrong model");
  PT_CHECK(B > 0 && W > 0, "pt_pq_quantize: empty input B=%d W=%d", B, W);
  const VQVAE& v = h->m.vq;
  hipStream_t s = (hipStream_t)stream;
  Arena ar(ws, ws_bytes);
  const int ld = ld_of(v.emb_dim);
  float* z_cl = ar.f32((size_t)B * W * ld);
  float* q_cl = ar.f32((size_t)B * W * ld);
  PT_CHECK(!ar.overflow, "pt_pq_quantize: workspace too small (%zu given, %zu needed)", ws_bytes, ar.off);
  if (launch_ncl_to_cl(z, z_cl, B, v.emb_dim, W, ld, 0, ld, s)) return 1;
  if (launch_pq_search(z_cl, ld, v.codebooks, B * W, v.groups, v.emb_dim / v.groups, v.num_codes, indices,
                       quantized ? q_cl : nullptr, ld, s)) return 1;
  if (quantized && launch_cl_to_ncl(q_cl, ld, 0, quantized, B, v.emb_dim, W, s)) return 1;
  return 0;
}
int pt_pq_quantize_static(pt_model* h, const float* z, int B, int W, float* quantized, int64_t* indices, void* stream) {
  PT_CHECK(h && z && indices, "pt_pq_quantize_static: null argument");
  PT_CHECK(h->m.finalized && (h->m.kind == "vqvae_sc" || h->m.kind == "vqvae_legacy"), "pt_pq_quantize_static: wrong model");
  PT_CHECK(B > 0 && W > 0, "pt_pq_quantize_static: empty input B=%d W=%d", B, W);
  const VQVAE& v = h->m.vq;
  const int ld = ld_of(v.emb_dim);
  const long long rows = (long long)B * W;
  PT_CHECK(rows <= PT_PQ_STATIC_MAX_ROWS && ld <= PT_PQ_STATIC_MAX_LD,
           "pt_pq_quantize_static: %lld latent rows of %d floats exceed the static scratch (at most %d rows of %d floats); "
           "use pt_pq_quantize with a caller workspace", rows, ld, PT_PQ_STATIC_MAX_ROWS, PT_PQ_STATIC_MAX_LD);
  // the address was resolved for the device the model was created on (pt_model_create), where it has to be used: no
  // lock and no runtime query here
  void* base = h->scratch;
  PT_CHECK(base != nullptr, "pt_pq_quantize_static: the model holds no static scratch address");
  hipStream_t s = (hipStream_t)stream;
  // both images are written in full (all ld columns of every row) before they are read: nothing has to start at zero
  float* z_cl = (float*)base;
  float* q_cl = z_cl + (size_t)PT_PQ_STATIC_MAX_ROWS * PT_PQ_STATIC_MAX_LD;
  if (launch_ncl_to_cl(z, z_cl, B, v.emb_dim, W, ld, 0, ld, s)) return 1;
  if (launch_pq_search(z_cl, ld, v.codebooks, B * W, v.groups, v.emb_dim / v.groups, v.num_codes, indices,
                       quantized ? q_cl : nullptr, ld, s, true)) return 1;
  if (quantized && launch_cl_to_ncl(q_cl, ld, 0, quantized, B, v.emb_dim, W, s)) return 1;
  return 0;
}
int pt_pq_quantize_all(pt_model* h, const void* idx, int idx_are_int64, int rows, float* out, void* stream) {
  PT_CHECK(h && idx && out, "pt_pq_quantize_all: null argument");
  PT_CHECK(h->m.finalized && (h->m.kind == "vqvae_sc" || h->m.kind == "vqvae_legacy"), "pt_pq_quantize_all: wrong model");
  const VQVAE& v = h->m.vq;
  return launch_pq_gather(idx, idx_are_int64, v.codebooks, rows, v.groups, v.emb_dim / v.groups, v.num_codes, out,
                          v.emb_dim, v.emb_dim, (hipStream_t)stream);
}

// ---- audio condition encoder
size_t pt_cond_encoder_workspace_bytes(const pt_model* h, int B, int T) {
  if (!h || !h->m.finalized) return 0;
  return cond_encoder_workspace_bytes(h->m, B, T);
}
int pt_cond_encoder_forward(pt_model* h, const float* aud, int B, int T, float* out, void* ws, size_t ws_bytes,
                            void* stream) {
  PT_CHECK(h && aud && out, "pt_cond_encoder_forward: null argument");
  return cond_encoder_forward(h->m, aud, B, T, out, nullptr, 0, 0, ws, ws_bytes, (hipStream_t)stream);
}

// ---- predictor / refiner
size_t pt_predictor_workspace_bytes(const pt_model* h, int B, int T) {
  if (!h || !h->m.finalized) return 0;
  return predictor_workspace_bytes(h->m, B, T);
}
int pt_predictor_condition(pt_model* h, const float* masked_motion, const float* mask, const float* audio, int B, int T,
                           float* condition, void* ws, size_t ws_bytes, void* stream) {
  PT_CHECK(h && masked_motion && mask && audio && condition, "pt_predictor_condition: null argument");
  return predictor_condition(h->m, masked_motion, mask, audio, B, T, condition, ws, ws_bytes, (hipStream_t)stream);
}
int pt_predictor_logits(pt_model* h, const int32_t* ids, const int64_t* label, const float* condition, int B, int Tq,
                        float* logits, void* ws, size_t ws_bytes, void* stream) {
  PT_CHECK(h && ids && label && condition && logits, "pt_predictor_logits: null argument");
  return predictor_logits(h->m, ids, label, condition, B, Tq, logits, ws, ws_bytes, (hipStream_t)stream);
}
int pt_predictor_sample(pt_model* h, const int64_t* label, const float* condition, const float* noise_q,
                        const float* noise_g, int steps, int B, int Tq, int32_t* codes_out, float* logits_out, void* ws,
                        size_t ws_bytes, void* stream) {
  PT_CHECK(h && label && condition && noise_q && noise_g && codes_out, "pt_predictor_sample: null argument");
  return predictor_sample(h->m, label, condition, noise_q, noise_g, steps, B, Tq, codes_out, logits_out, ws, ws_bytes,
                          (hipStream_t)stream);
}
int pt_predictor_sample_seeded(pt_model* h, const int64_t* label, const float* condition, unsigned long long seed, int steps, int B,
                               int Tq, int32_t* codes_out, float* logits_out, void* ws, size_t ws_bytes, void* stream) {
  PT_CHECK(h && label && condition && codes_out, "pt_predictor_sample_seeded: null argument");
  return predictor_sample(h->m, label, condition, nullptr, nullptr, steps, B, Tq, codes_out, logits_out, ws, ws_bytes,
                          (hipStream_t)stream, seed);
}
int pt_predictor_sample_seed_dev(pt_model* h, const int64_t* label, const float* condition, const unsigned long long* seed_dev,
                                 int steps, int B, int Tq, int32_t* codes_out, float* logits_out, void* ws, size_t ws_bytes,
                                 void* stream) {
  PT_CHECK(h && label && condition && codes_out && seed_dev, "pt_predictor_sample_seed_dev: null argument");
  return predictor_sample(h->m, label, condition, nullptr, nullptr, steps, B, Tq, codes_out, logits_out, ws, ws_bytes,
                          (hipStream_t)stream, 0, seed_dev);
}
int pt_maskgit_noise(unsigned long long seed, int step, int B, int N, int V, float* noise_q, float* noise_g, void* stream) {
  return launch_maskgit_noise(seed, step, B, N, V, noise_q, noise_g, (hipStream_t)stream);
}
int pt_maskgit_noise_from_bits(const uint32_t* bits, int n, float* u, float* noise_q, float* noise_g, void* stream) {
  return launch_maskgit_bits(bits, n, u, noise_q, noise_g, (hipStream_t)stream);
}
size_t pt_refiner_workspace_bytes(const pt_model* h, int B, int T) {
  if (!h || !h->m.finalized) return 0;
  return refiner_workspace_bytes(h->m, B, T);
}
int pt_refiner_forward(pt_model* h, const float* x, const float* condition, const float* mask, const int64_t* label,
                       int n_label, int B, int T, float* out, void* ws, size_t ws_bytes, void* stream) {
  PT_CHECK(h && x && condition && mask && label && out, "pt_refiner_forward: null argument");
  return refiner_forward(h->m, x, condition, mask, label, n_label, B, T, out, ws, ws_bytes, (hipStream_t)stream);
}
int pt_interp_blend(const float* pred, int B, int C, int T_in, int T_out, const float* gt, const float* mask,
                    int premask, float* out, void* stream) {
  PT_CHECK(out && (pred || gt), "pt_interp_blend: null argument");
  return launch_interp_blend(pred, B, C, T_in, T_out, gt, mask, premask, out, (hipStream_t)stream);
}

// ---- op level
int pt_conv1d_cl(const float* A, int lda, const float* Wpacked, const float* bias, int N, int Npad, int K, int kind,
                 int B, int Lin, float* C, int ldc, int act, float slope, const float* Res, int ldr, void* stream) {
  PT_CHECK(A && Wpacked && C, "pt_conv1d_cl: null argument");
  PT_CHECK(kind >= 0 && kind <= 4, "pt_conv1d_cl: kind %d out of range", kind);
  DevConv c;
  c.w = const_cast<float*>(Wpacked);
  c.b = const_cast<float*>(bias);
  c.N = N; c.Npad = Npad; c.K = K; c.Cin = K;
  static const int slabs[] = {1, 3, 7, 4, 4};
  c.slabs = slabs[kind];
  return run_conv(c, (ConvKind)kind, A, lda, B, Lin, C, ldc, act, slope, Res, ldr, (hipStream_t)stream);
}
int pt_conv1d_cl_bf16(const void* A, int lda, const void* Wpacked, const float* bias, int N, int Npad, int K, int kind, int B,
                      int Lin, float* C, int ldc, void* Cb, int ldcb, int act, float slope, const void* ResB, int ldr,
                      void* stream) {
  PT_CHECK(A && Wpacked && (C || Cb), "pt_conv1d_cl_bf16: null argument");
  PT_CHECK(kind >= 0 && kind <= 4, "pt_conv1d_cl_bf16: kind %d out of range", kind);
  if (conv8_prepare()) return 1;
  DevConv c;
  c.wb = (unsigned short*)const_cast<void*>(Wpacked);
  c.b = const_cast<float*>(bias);
  c.N = N; c.Npad = Npad; c.K = K; c.Kb = K; c.Cin = K;
  static const int slabs[] = {1, 3, 7, 4, 4};
  c.slabs = slabs[kind];
  return run_conv_b(c, (ConvKind)kind, (const bf16_t*)A, lda, B, Lin, C, ldc, (bf16_t*)Cb, ldcb, act, slope, (const bf16_t*)ResB,
                    ldr, (hipStream_t)stream, nullptr, -1);
}
int pt_ncl_to_cl(const float* in, float* out, int B, int C, int L, int ldo, void* stream) {
  PT_CHECK(in && out, "pt_ncl_to_cl: null argument");
  return launch_ncl_to_cl(in, out, B, C, L, ldo, 0, ldo, (hipStream_t)stream);
}
int pt_cl_to_ncl(const float* in, int ldi, float* out, int B, int C, int L, void* stream) {
  PT_CHECK(in && out, "pt_cl_to_ncl: null argument");
  return launch_cl_to_ncl(in, ldi, 0, out, B, C, L, (hipStream_t)stream);
}
int pt_pq_search_cl(const float* z, int ldz, const float* codebooks, int rows, int G, int D, int K, int64_t* idx,
                    float* q, int ldq, void* stream) {
  PT_CHECK(z && codebooks && idx, "pt_pq_search_cl: null argument");
  return launch_pq_search(z, ldz, codebooks, rows, G, D, K, idx, q, ldq, (hipStream_t)stream);
}

int pt_prof_enable(int on) {
  prof_enable(on != 0);
  return 0;
}
int pt_prof_read(double* ms8, double* flops8, int* launches8) {
  PT_CHECK(ms8 && flops8 && launches8, "pt_prof_read: null argument");
  return prof_read(ms8, flops8, launches8);
}
int pt_gemm_bf16(const void* A, int lda, const void* W, const float* bias, int M, int N, int K, int Npad, float* C,
                 int ldc, void* Cb, int ldcb, int relu, void* stream) {
  PT_CHECK(A && W && (C || Cb), "pt_gemm_bf16: null argument");
  if (N >= 256 && N % 8 == 0) {   // same dispatch as the transformer sequences (transformer.hip lin_b)
    GemmBigArgs b;
    memset(&b, 0, sizeof(b));
    b.A = (const unsigned short*)A; b.lda = lda; b.W = (const unsigned short*)W; b.bias = bias;
    b.M = M; b.N = N; b.K = K; b.Npad = Npad; b.epi = EPI_STD;
    b.C = C; b.ldc = ldc; b.Cb = (unsigned short*)Cb; b.ldcb = ldcb; b.relu = relu;
    return launch_gemm_big(b, (hipStream_t)stream);
  }
  GemmArgsB a;
  memset(&a, 0, sizeof(a));
  a.A = (const unsigned short*)A; a.lda = lda; a.W = (const unsigned short*)W; a.bias = bias;
  a.C = C; a.ldc = ldc; a.Cb = (unsigned short*)Cb; a.ldcb = ldcb;
  a.M = M; a.N = N; a.K = K; a.Npad = Npad; a.ntaps = 1;
  a.Lgrid = M; a.Lin = M; a.Lout = M; a.in_stride = 1; a.out_stride = 1;
  a.act = relu ? ACT_LEAKY : ACT_NONE; a.slope = 0.f;
  return launch_gemm_bf16(a, (hipStream_t)stream);
}
int pt_gemm_bf16_lnmod(const void* A, int lda, const void* W, const float* bias, int M, int K, int Npad,
                       const float* style_folded, const int64_t* label, int n_label, int rows_per_clip, int n_classes,
                       void* out_bf16, void* stream) {
  PT_CHECK(A && W && out_bf16, "pt_gemm_bf16_lnmod: null argument");
  GemmBigArgs b;
  memset(&b, 0, sizeof(b));
  b.A = (const unsigned short*)A; b.lda = lda; b.W = (const unsigned short*)W; b.bias = bias;
  b.M = M; b.N = 512; b.K = K; b.Npad = Npad; b.epi = EPI_LNMOD;
  b.Cb = (unsigned short*)out_bf16; b.ldcb = 512; b.style = style_folded;
  b.label = label; b.n_label = n_label; b.rows_per_clip = rows_per_clip; b.n_classes = n_classes;
  return launch_gemm_big(b, (hipStream_t)stream);
}
int pt_gemm_bf16_addln(const void* A, int lda, const void* W, const float* bias, int M, int K, int Npad,
                       const float* gamma, const float* beta, float* x, void* x_bf16, void* stream) {
  PT_CHECK(A && W && x_bf16, "pt_gemm_bf16_addln: null argument");
  GemmBigArgs b;
  memset(&b, 0, sizeof(b));
  b.A = (const unsigned short*)A; b.lda = lda; b.W = (const unsigned short*)W; b.bias = bias;
  b.M = M; b.N = 512; b.K = K; b.Npad = Npad; b.epi = EPI_ADDLN;
  b.Cb = (unsigned short*)x_bf16; b.ldcb = 512; b.gamma = gamma; b.beta = beta; b.X = x;
  return launch_gemm_big(b, (hipStream_t)stream);
}
int pt_postprocess_poses(const float* pred, int B, int T, const float* mean, const float* std_, const int* c_index,
                         int n_stats, const float* lower_pose, float* out, int ld_out, void* stream) {
  return launch_postprocess(pred, B, T, mean, std_, c_index, n_stats, lower_pose, out, ld_out, (hipStream_t)stream);
}
int pt_permute_k32(const void* W, void* out, int N, int K, void* stream) {
  return launch_permute_k32((const unsigned short*)W, (unsigned short*)out, N, K, (hipStream_t)stream);
}
int pt_gemm_bf16_sublayer(const void* A, int lda, int K1, const void* W1, const float* b1, const float* style_folded,
                          const int64_t* label, int n_label, int rows_per_clip, int n_classes, const void* W2p, const float* b2,
                          const float* gamma, const float* beta, void* x_bf16, int M, void* stream) {
  SubArgs a;
  memset(&a, 0, sizeof(a));
  a.A = (const unsigned short*)A; a.lda = lda; a.K1 = K1; a.W1 = (const unsigned short*)W1; a.b1 = b1;
  a.fold = style_folded; a.label = label; a.n_label = n_label; a.rows_per_clip = rows_per_clip; a.n_classes = n_classes;
  a.W2p = (const unsigned short*)W2p; a.b2 = b2; a.gamma = gamma; a.beta = beta; a.X = (unsigned short*)x_bf16;
  a.M = M;
  return launch_sublayer(a, (hipStream_t)stream);
}
int pt_attention_cl(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, int B,
                    int Nq, int Nk, int H, void* stream) {
  PT_CHECK(Q && K && V && O, "pt_attention_cl: null argument");
  return launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, B, Nq, Nk, H, (hipStream_t)stream);
}
int pt_attention_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int B,
                      int Nq, int Nk, int H, void* stream) {
  PT_CHECK(Q && K && V && O, "pt_attention_bf16: null argument");
  return launch_attention_bf16((const unsigned short*)Q, ldq, (const unsigned short*)K, ldk, (const unsigned short*)V, ldv,
                               (unsigned short*)O, ldo, B, Nq, Nk, H, (hipStream_t)stream);
}
int pt_pq_gather_cl(const void* idx, int idx_are_int64, const float* codebooks, int rows, int G, int D, int K,
                    float* out, int ldq, void* stream) {
  PT_CHECK(idx && codebooks && out, "pt_pq_gather_cl: null argument");
  return launch_pq_gather(idx, idx_are_int64, codebooks, rows, G, D, K, out, ldq, G * D, (hipStream_t)stream);
}

/* ---- training-mode op-level entry points (train.hip) ---- */
int pt_embed_pe(const int32_t* ids, const float* tok, const double* pe, const double* ppe, float* x, int rows, int N, int G,
                int vocab, void* stream) {
  PT_CHECK(ids && tok && pe && ppe && x && rows > 0 && N > 0 && G > 0, "pt_embed_pe: bad argument");
  return launch_embed_pe(ids, tok, pe, ppe, x, nullptr, rows, N, G, vocab, (hipStream_t)stream);
}
int pt_train_wgrad(const float* dY, int lddy, const float* X, int ldx, float* dW, int N, int Cin, int kind, int B, int Lin,
                   void* stream) { return launch_wgrad(dY, lddy, X, ldx, dW, nullptr, N, Cin, kind, B, Lin, (hipStream_t)stream); }
int pt_train_wgrad_bias(const float* dY, int lddy, const float* X, int ldx, float* dW, float* db, int N, int Cin, int kind, int B,
                        int Lin, void* stream) {
  PT_CHECK(db != nullptr, "pt_train_wgrad_bias: db is null (use pt_train_wgrad)");
  return launch_wgrad(dY, lddy, X, ldx, dW, db, N, Cin, kind, B, Lin, (hipStream_t)stream);
}
int pt_train_colsum(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* out, void* stream) {
  return launch_colsum(a, lda, b, ldb, rows, cols, out, (hipStream_t)stream);
}
int pt_train_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y, float* stats, int rows,
                    void* stream) { return launch_ln_fwd(x, res, gamma, beta, y, stats, rows, (hipStream_t)stream); }
int pt_train_ln_bwd(const float* dy, const float* x, const float* res, const float* stats, const float* gamma, float* dx,
                    float* dgamma, float* dbeta, int rows, void* stream) {
  return launch_ln_bwd(dy, x, res, stats, gamma, dx, dgamma, dbeta, rows, (hipStream_t)stream);
}
int pt_train_modsilu_fwd(const float* y, const float* table, const int64_t* label, int n_label, int rows_per_clip, int n_classes,
                         float* h, int rows, void* stream) {
  return launch_modsilu_fwd(y, table, label, n_label, rows_per_clip, n_classes, h, rows, (hipStream_t)stream);
}
int pt_train_modsilu_bwd(const float* dh, const float* y, const float* table, const int64_t* label, int n_label,
                         int rows_per_clip, int n_classes, float* dy, float* dtable, int rows, void* stream) {
  return launch_modsilu_bwd(dh, y, table, label, n_label, rows_per_clip, n_classes, dy, dtable, rows, (hipStream_t)stream);
}
int pt_train_attention_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* dO, int ldo,
                           float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, float* lse, float* delta, int B,
                           int Nq, int Nk, int H, float pdrop, uint64_t seed, void* stream) {
  return launch_attention_bwd(Q, ldq, K, ldk, V, ldv, dO, ldo, dQ, lddq, dK, lddk, dV, lddv, lse, delta, B, Nq, Nk, H, pdrop,
                              seed, (hipStream_t)stream);
}
int pt_train_attention_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, int B,
                           int Nq, int Nk, int H, float pdrop, uint64_t seed, void* stream) {
  return launch_attention_train_fwd(Q, ldq, K, ldk, V, ldv, O, ldo, B, Nq, Nk, H, pdrop, seed, (hipStream_t)stream);
}
int pt_train_cross_entropy(const float* logits, const int64_t* target, int V, int rows, float* loss, float* dlogits,
                           void* stream) { return launch_ce(logits, target, V, rows, loss, dlogits, (hipStream_t)stream); }
int pt_train_bn_fwd(const float* x, int ldx, int rows, int cols, const float* gamma, const float* beta, const float* res, int ldr,
                    float slope, float* y, int ldy, float* mean_rstd, double* acc_ws, float* run_mean, float* run_var,
                    float momentum, void* stream) {
  return launch_bn_train_fwd(x, ldx, rows, cols, gamma, beta, res, ldr, slope, y, ldy, mean_rstd, acc_ws, run_mean, run_var,
                             momentum, (hipStream_t)stream);
}
int pt_train_bn_bwd(const float* dy, int lddy, const float* y, int ldy, float slope, const float* x, int ldx,
                    const float* mean_rstd, const float* gamma, float* dz, int lddz, float* dx, int lddx, float* sums, int rows,
                    int cols, void* stream) {
  return launch_bn_train_bwd(dy, lddy, y, ldy, slope, x, ldx, mean_rstd, gamma, dz, lddz, dx, lddx, sums, rows, cols,
                             (hipStream_t)stream);
}
int pt_train_act_bwd(const float* dy, int lddy, const float* y, int ldy, float slope, float* dz, int lddz, int rows, int cols,
                     void* stream) { return launch_act_bwd(dy, lddy, y, ldy, slope, dz, lddz, rows, cols, (hipStream_t)stream); }
int pt_train_gate_fwd(const float* hv, int rows, int C, float* out, void* stream) {
  return launch_gate_fwd(hv, rows, C, out, (hipStream_t)stream);
}
int pt_train_gate_bwd(const float* dout, const float* hv, int rows, int C, float* dhv, void* stream) {
  return launch_gate_bwd(dout, hv, rows, C, dhv, (hipStream_t)stream);
}
int pt_train_embed_bwd(const float* dx, const int32_t* ids, int rows, int vocab, float* dE, void* stream) {
  return launch_embed_bwd(dx, ids, rows, vocab, dE, (hipStream_t)stream);
}
int pt_train_dropout(const float* x, float* y, size_t n, float p, uint64_t seed, void* stream) {
  return launch_dropout(x, y, n, p, seed, (hipStream_t)stream);
}
int pt_train_sumsq(const float* g, size_t n, float* out, void* stream) { return launch_sumsq(g, n, out, (hipStream_t)stream); }
int pt_train_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream) {
  return launch_clip_coef(sumsq, max_norm, coef, norm_out, (hipStream_t)stream);
}
int pt_train_adamw(float* p, const float* g, float* m, float* v, size_t n, const float* gscale, float lr, float beta1,
                   float beta2, float eps, float wd, int step, void* stream) {
  return launch_adamw(p, g, m, v, n, gscale, lr, beta1, beta2, eps, wd, step, (hipStream_t)stream);
}
int pt_train_add(const float* a, const float* b, float* out, size_t n, void* stream) {
  return launch_add(a, b, out, n, (hipStream_t)stream);
}
int pt_train_pq_ema(const float* z, int ldz, const int64_t* idx, int rows, int G, int D, int K, float* counts, float* dw,
                    float* cs_hidden, float* dw_hidden, float* emb, float decay, float eps, int counter, const float* tile_noise,
                    uint64_t tile_seed, void* stream) {
  return launch_pq_ema(z, ldz, idx, rows, G, D, K, counts, dw, cs_hidden, dw_hidden, emb, decay, eps, counter, (hipStream_t)stream, 1,
                       tile_noise, tile_seed);
}
int pt_train_pq_ema_shared(const float* z, int ldz, const int64_t* idx, int rows, int G, int D, int K, float* counts, float* dw,
                           float* cs_hidden, float* dw_hidden, float* emb, float decay, float eps, int counter, const float* tile_noise,
                           uint64_t tile_seed, void* stream) {
  return launch_pq_ema(z, ldz, idx, rows, G, D, K, counts, dw, cs_hidden, dw_hidden, emb, decay, eps, counter, (hipStream_t)stream, 2,
                       tile_noise, tile_seed);
}
int pt_train_vq_ema(const float* z, int ldz, const int64_t* idx, int rows, int D, int K, float* counts, float* dw,
                    float* cs_hidden, float* dw_hidden, float* emb, float decay, float eps, int counter, void* stream) {
  return launch_pq_ema(z, ldz, idx, rows, 1, D, K, counts, dw, cs_hidden, dw_hidden, emb, decay, eps, counter, (hipStream_t)stream, 0,
                       nullptr, 0);
}
int pt_train_vq_losses(const float* pred, int ldp, const float* gt, int ldg, int B, int T, int nparts, const int* bounds,
                       const float* weights, int vel_cols, float beta, float* sums, float* dpred, int lddp, void* stream) {
  return launch_vq_losses(pred, ldp, gt, ldg, B, T, nparts, bounds, weights, vel_cols, beta, sums, dpred, lddp, (hipStream_t)stream,
                          nullptr, 0.f, 0);
}
int pt_train_masked_l1_losses(const float* pred, int ldp, const float* gt, int ldg, const float* nmask, float t_nmask, int B, int T,
                              int nparts, const int* bounds, const float* weights, int vel_cols, float* sums, float* dpred,
                              int lddp, void* stream) {
  return launch_vq_losses(pred, ldp, gt, ldg, B, T, nparts, bounds, weights, vel_cols, 0.f, sums, dpred, lddp, (hipStream_t)stream,
                          nmask, t_nmask, 1);
}
int pt_train_sqdiff(const float* a, const float* b, size_t n, float* out, void* stream) {
  return launch_sqdiff(a, b, n, out, (hipStream_t)stream);
}
int pt_train_commit_bwd(const float* dq, const float* z, const float* q, const float* g, float coef, size_t n, float* dz,
                        void* stream) {
  return launch_commit_bwd(dq, z, q, g, coef, n, dz, (hipStream_t)stream);
}
}  // extern "C"
