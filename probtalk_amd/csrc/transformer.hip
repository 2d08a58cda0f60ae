// Predictor (motion-context gated convs + MaskGIT PTransformer) and refiner (RTransformer) module
// sequences.  Reference structure followed (eval mode):
//   nets/inpainting/predictornet.py:31-112, nets/inpainting/gated_pixelcnn_1d.py:39-83,
//   nets/inpainting/PTransformer.py:134-140 (style 'sb'), 314-372 (decoder layer), 590-704 (MaskGIT),
//   nets/inpainting/RTransformer.py:181-230.
// The cross-attention K/V of every layer depend only on the condition, so the MaskGIT loop computes
// them once and reuses them in all passes.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "transformer.h"

namespace pt {

int launch_embed_pe(const int* ids, const float* tok, const double* pe, const double* ppe, float* x,
                    unsigned short* xb, int rows, int N, int G, int vocab, hipStream_t s);
int launch_ln_mod_silu(const float* h, float* out, unsigned short* outb, const float* gamma, const float* beta,
                       const float* style, const int64_t* label, int n_label, int rows, int rows_per_clip,
                       int n_classes, hipStream_t s);
int launch_add_ln_b(unsigned short* xb, const float* y, const float* gamma, const float* beta, int rows, hipStream_t s);
int launch_add_ln(float* x, const float* y, const float* gamma, const float* beta, int rows, unsigned short* xb,
                  hipStream_t s);
int launch_cvt_f32_bf16(const float* in, int ldi, unsigned short* out, int ldo, int rows, int cols, int zero_to,
                        hipStream_t s);
int launch_attention_bf16(const unsigned short* Q, int ldq, const unsigned short* K, int ldk, const unsigned short* V,
                          int ldv, unsigned short* O, int ldo, int B, int Nq, int Nk, int H, hipStream_t s);
int launch_mask_onehot(const float* mask, float* cond, int rows, int ld, int c0, int zero_to, hipStream_t s);
int launch_maskgit_step(const float* logits, const float* noise_q, const float* noise_g, int* ids, int B, int N, int V,
                        int mask_id, int want_mask, float temp, hipStream_t s, unsigned long long seed, int step,
                        const unsigned long long* seed_dev);
int launch_fill_i32(int* p, int v, size_t n, hipStream_t s);
int launch_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                     int B, int Nq, int Nk, int H, hipStream_t s);

constexpr int DM = 512, NH = 8, FF = 2048;

struct LNW { float* g = nullptr; float* b = nullptr; };
struct StyleW {
  float* table = nullptr;   // [n_classes][1024]: scale | shift = emb_layers_2(SiLU(class_embed[c]))
  float* fold = nullptr;    // [n_classes][1024]: gamma*(1+scale) | beta*(1+scale)+shift (LayerNorm affine folded in; bf16 mode)
  LNW ln; DevConv out;
  unsigned short* out_wp = nullptr;   // bf16 mode: out.wb with the K axis permuted for the fused tail (launch_permute_k32)
};
struct DecLayer {
  DevConv sa_in, sa_out, ca_q, ca_kv, ca_out, ff1, ff2;
  LNW n1, n2, n3;
  StyleW st[3];
};
struct GatedW { DevConv vert, res, skip; ConvKind kind; int dim, in_dim; bool strided; };

struct Predictor {
  int mot_dim, num_codes, groups, n_classes, n_layers, audio_dim;   // audio_dim: audio (+ text) feature channels
  int cond_w = 768;       // condition width = motion 512 + audio_dim (768; 1024 with text features)
  DevConv motion_emb, condi_emb, out0, out2;
  GatedW enc[10];
  float* tok_emb = nullptr;
  double *pe = nullptr, *ppe = nullptr;
  int pe_rows = 0;
  std::vector<DecLayer> layers;
};
struct Refiner {
  int in_dim, cond_dim, n_classes, n_layers;
  DevConv input_conv, audio_conv, out0, out2;
  double* pe = nullptr;
  int pe_rows = 0;
  std::vector<DecLayer> layers;
};

// ------------------------------------------------------------------ packing helpers (runtime.hip has the conv packer)
int pack_conv_public(Model& m, DevConv& out, const std::string& w1, const std::string& bn, const std::string& w2,
                     bool transposed, bool gated);

static const HostTensor* findt(const Model& m, const std::string& name) {
  auto it = m.sd.find(name);
  return it == m.sd.end() ? nullptr : &it->second;
}
template <typename T>
static int upload_v(Model& m, const std::vector<T>& h, T** dev) {
  void* p = nullptr;
  PT_HIP(hipMalloc(&p, h.size() * sizeof(T)));
  m.allocs.push_back(p);
  PT_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *dev = (T*)p;
  return 0;
}
static int upload_named(Model& m, const std::string& name, float** dev, int64_t expect) {
  const HostTensor* t = findt(m, name);
  PT_CHECK(t != nullptr && t->dtype == 0, "missing fp32 tensor %s", name.c_str());
  PT_CHECK(expect < 0 || t->numel() == expect, "%s: %lld elements where %lld expected", name.c_str(),
           (long long)t->numel(), (long long)expect);
  return upload_v(m, t->f, dev);
}
static int upload_ln(Model& m, const std::string& p, LNW& ln) {
  if (upload_named(m, p + ".weight", &ln.g, DM)) return 1;
  return upload_named(m, p + ".bias", &ln.b, DM);
}
// rows [r0, r0+n) of a (rows, cols) matrix + its bias become tensors "<alias>.weight/.bias"
static int alias_rows(Model& m, const std::string& w, const std::string& b, int r0, int n, const std::string& alias) {
  const HostTensor *W = findt(m, w), *Bv = findt(m, b);
  PT_CHECK(W && Bv && W->shape.size() == 2, "missing projection %s / %s", w.c_str(), b.c_str());
  const int cols = (int)W->shape[1];
  PT_CHECK(r0 + n <= W->shape[0], "%s: rows %d..%d out of range", w.c_str(), r0, r0 + n);
  HostTensor tw, tb;
  tw.shape = {n, cols};
  tw.f.assign(W->f.begin() + (size_t)r0 * cols, W->f.begin() + (size_t)(r0 + n) * cols);
  tb.shape = {n};
  tb.f.assign(Bv->f.begin() + r0, Bv->f.begin() + r0 + n);
  m.sd[alias + ".weight"] = std::move(tw);
  m.sd[alias + ".bias"] = std::move(tb);
  return 0;
}

// style table[c] = emb_layers_2.1(SiLU(class_embed[c]))  (PTransformer.py:135-136): depends on weights only
static int pack_style(Model& m, const std::string& p, const std::string& class_embed, int n_classes, StyleW& st) {
  const HostTensor *ce = findt(m, class_embed), *W = findt(m, p + "emb_layers_2.1.weight"),
                   *Bv = findt(m, p + "emb_layers_2.1.bias");
  PT_CHECK(ce && W && Bv, "missing style tensors under %s", p.c_str());
  PT_CHECK(ce->shape.size() == 2 && ce->shape[0] == n_classes && ce->shape[1] == DM, "%s: expected (%d,%d)",
           class_embed.c_str(), n_classes, DM);
  PT_CHECK(W->numel() == 2 * DM * DM, "%semb_layers_2.1.weight: expected (%d,%d)", p.c_str(), 2 * DM, DM);
  std::vector<float> table((size_t)n_classes * 2 * DM);
  std::vector<float> e(DM);
  for (int c = 0; c < n_classes; ++c) {
    for (int k = 0; k < DM; ++k) {
      const float x = ce->f[(size_t)c * DM + k];
      e[k] = x / (1.f + expf(-x));
    }
    for (int n = 0; n < 2 * DM; ++n) {
      double acc = 0.0;
      for (int k = 0; k < DM; ++k) acc += (double)W->f[(size_t)n * DM + k] * (double)e[k];
      table[(size_t)c * 2 * DM + n] = (float)(acc + (double)Bv->f[n]);
    }
  }
  if (upload_v(m, table, &st.table)) return 1;
  if (upload_ln(m, p + "norm_2", st.ln)) return 1;
  {
    const HostTensor *g = findt(m, p + "norm_2.weight"), *b = findt(m, p + "norm_2.bias");
    std::vector<float> fold((size_t)n_classes * 2 * DM);
    for (int c = 0; c < n_classes; ++c)
      for (int n = 0; n < DM; ++n) {
        const float sc = table[(size_t)c * 2 * DM + n], sh = table[(size_t)c * 2 * DM + DM + n];
        fold[(size_t)c * 2 * DM + n] = g->f[n] * (1.f + sc);
        fold[(size_t)c * 2 * DM + DM + n] = b->f[n] * (1.f + sc) + sh;
      }
    if (upload_v(m, fold, &st.fold)) return 1;
  }
  if (pack_conv_public(m, st.out, p + "out_layers_2.2", "", "", false, false)) return 1;
  if (st.out.wb) {
    PT_CHECK(st.out.Npad == DM && st.out.Kb == DM, "%sout_layers_2.2: expected a (%d,%d) Linear", p.c_str(), DM, DM);
    void* wp = nullptr;
    PT_HIP(hipMalloc(&wp, (size_t)DM * DM * 2));
    m.allocs.push_back(wp);
    st.out_wp = (unsigned short*)wp;
    if (launch_permute_k32(st.out.wb, st.out_wp, DM, DM, nullptr)) return 1;
    PT_HIP(hipStreamSynchronize(nullptr));
  }
  return 0;
}

static int pack_layer(Model& m, const std::string& p, const std::string& class_embed, int n_classes, DecLayer& L) {
  if (alias_rows(m, p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", 0, 3 * DM, p + "self_attn.in_proj")) return 1;
  if (alias_rows(m, p + "multihead_attn.in_proj_weight", p + "multihead_attn.in_proj_bias", 0, DM, p + "multihead_attn.q")) return 1;
  if (alias_rows(m, p + "multihead_attn.in_proj_weight", p + "multihead_attn.in_proj_bias", DM, 2 * DM, p + "multihead_attn.kv")) return 1;
  if (pack_conv_public(m, L.sa_in, p + "self_attn.in_proj", "", "", false, false)) return 1;
  if (pack_conv_public(m, L.sa_out, p + "self_attn.out_proj", "", "", false, false)) return 1;
  if (pack_conv_public(m, L.ca_q, p + "multihead_attn.q", "", "", false, false)) return 1;
  if (pack_conv_public(m, L.ca_kv, p + "multihead_attn.kv", "", "", false, false)) return 1;
  if (pack_conv_public(m, L.ca_out, p + "multihead_attn.out_proj", "", "", false, false)) return 1;
  if (pack_conv_public(m, L.ff1, p + "linear1", "", "", false, false)) return 1;
  if (pack_conv_public(m, L.ff2, p + "linear2", "", "", false, false)) return 1;
  PT_CHECK(L.sa_in.N == 3 * DM && L.sa_in.Cin == DM && L.ff1.N == FF && L.ff2.Cin == FF,
           "%s: unexpected transformer layer geometry", p.c_str());
  if (upload_ln(m, p + "norm1", L.n1) || upload_ln(m, p + "norm2", L.n2) || upload_ln(m, p + "norm3", L.n3)) return 1;
  for (int i = 0; i < 3; ++i)
    if (pack_style(m, p + "style_gourp_" + std::to_string(i + 1) + ".", class_embed, n_classes, L.st[i])) return 1;
  return 0;
}

static int upload_pe(Model& m, const std::string& name, double** dev, int* rows) {
  const HostTensor* t = findt(m, name);
  PT_CHECK(t != nullptr && t->dtype == 1 && t->shape.size() == 2 && t->shape[1] == DM,
           "%s: expected a float64 (rows,%d) positional table", name.c_str(), DM);
  *rows = (int)t->shape[0];
  return upload_v(m, t->d, dev);
}

static int finalize_predictor(Model& m) {
  PT_CHECK(m.cfg.size() >= 6, "predictor: cfg = {mot_dim, num_codes, groups, n_classes, n_layers, audio_dim}");
  Predictor* P = new Predictor();
  m.pred = P;
  P->mot_dim = m.cfg[0]; P->num_codes = m.cfg[1]; P->groups = m.cfg[2]; P->n_classes = m.cfg[3];
  P->n_layers = m.cfg[4]; P->audio_dim = m.cfg[5];
  if (pack_conv_public(m, P->motion_emb, "motion_embedding", "", "", false, false)) return 1;
  PT_CHECK(P->motion_emb.Cin == P->mot_dim + 1, "motion_embedding: expected %d input channels", P->mot_dim + 1);
  int d = P->motion_emb.N;
  for (int i = 0; i < 10; ++i) {
    GatedW& g = P->enc[i];
    const std::string p = "enc." + std::to_string(i) + ".";
    g.strided = (i == 2 || i == 4 || i == 6);
    g.kind = i == 0 ? CONV_K7 : (g.strided ? CONV_DOWN : CONV_K3);
    g.in_dim = d;
    if (i == 4 || i == 6) d *= 2;
    g.dim = d;
    if (pack_conv_public(m, g.vert, p + "vert_stack", "", "", false, true)) return 1;
    if (pack_conv_public(m, g.res, p + "res", "", "", false, false)) return 1;
    if (g.strided && pack_conv_public(m, g.skip, p + "conv", "", "", false, false)) return 1;
    PT_CHECK(g.vert.N == 2 * g.dim && g.vert.Cin == g.in_dim, "%svert_stack: unexpected geometry", p.c_str());
  }
  if (upload_named(m, "decoder.tok_emb.weight", &P->tok_emb, (int64_t)(P->num_codes + 1) * DM)) return 1;
  if (upload_pe(m, "decoder.PE.position_encoding.weight", &P->pe, &P->pe_rows)) return 1;
  int ppe_rows;
  if (upload_pe(m, "decoder.product_PE.position_encoding.weight", &P->ppe, &ppe_rows)) return 1;
  PT_CHECK(ppe_rows == P->groups + 1, "product_PE has %d rows, groups+1 = %d expected", ppe_rows, P->groups + 1);
  if (pack_conv_public(m, P->condi_emb, "decoder.condi_emb", "", "", false, false)) return 1;
  PT_CHECK(P->condi_emb.Cin == d + P->audio_dim, "condi_emb expects %d channels, motion %d + audio %d given",
           P->condi_emb.Cin, d, P->audio_dim);
  P->cond_w = d + P->audio_dim;
  PT_CHECK(P->cond_w % 64 == 0 && P->audio_dim > 0, "predictor: condition width %d must be a multiple of 64", P->cond_w);
  if (pack_conv_public(m, P->out0, "decoder.output_conv.0", "", "", false, false)) return 1;
  if (pack_conv_public(m, P->out2, "decoder.output_conv.2", "", "", false, false)) return 1;
  PT_CHECK(P->out2.N == P->num_codes, "output head has %d classes, %d codes expected", P->out2.N, P->num_codes);
  P->layers.resize(P->n_layers);
  for (int i = 0; i < P->n_layers; ++i)
    if (pack_layer(m, "decoder.transformer_decoder.layers." + std::to_string(i) + ".", "decoder.class_embed.weight",
                   P->n_classes, P->layers[i])) return 1;
  return 0;
}

static int finalize_refiner(Model& m) {
  PT_CHECK(m.cfg.size() >= 4, "refiner: cfg = {in_dim, cond_dim, n_classes, n_layers}");
  Refiner* R = new Refiner();
  m.ref = R;
  R->in_dim = m.cfg[0]; R->cond_dim = m.cfg[1]; R->n_classes = m.cfg[2]; R->n_layers = m.cfg[3];
  if (pack_conv_public(m, R->input_conv, "input_conv", "", "", false, false)) return 1;
  if (pack_conv_public(m, R->audio_conv, "audio_conv", "", "", false, false)) return 1;
  PT_CHECK(R->input_conv.Cin == R->in_dim && R->audio_conv.Cin == R->cond_dim + 2,
           "refiner: input_conv/audio_conv geometry (motion_context adds 2 mask channels)");
  if (pack_conv_public(m, R->out0, "output_conv.0", "", "", false, false)) return 1;
  if (pack_conv_public(m, R->out2, "output_conv.2", "", "", false, false)) return 1;
  if (upload_pe(m, "PE.position_encoding.weight", &R->pe, &R->pe_rows)) return 1;
  R->layers.resize(R->n_layers);
  for (int i = 0; i < R->n_layers; ++i)
    if (pack_layer(m, "transformer_decoder.layers." + std::to_string(i) + ".", "class_embed.weight", R->n_classes,
                   R->layers[i])) return 1;
  return 0;
}

void free_transformer_models(Model& m) {
  delete m.pred;
  delete m.ref;
  m.pred = nullptr;
  m.ref = nullptr;
}

int finalize_transformer_models(Model& m) {
  if (m.kind == "predictor") return finalize_predictor(m);
  if (m.kind == "refiner") return finalize_refiner(m);
  set_error("unknown model kind %s", m.kind.c_str());
  return 1;
}

// ------------------------------------------------------------------ decoder layer
// Buffers of one transformer stack.  fp32 mode uses x/qkv/att/t/hb as floats; bf16 mode keeps the residual
// stream x and the pre-LayerNorm sub-layer outputs t in fp32 and everything that only feeds a GEMM or the
// attention core (xb, qkv, att, hb) in bf16.
struct TBufs {
  float *x = nullptr, *qkv = nullptr, *att = nullptr, *t = nullptr, *hb = nullptr;   // fp32 mode
  bf16_t *xb = nullptr, *qkvb = nullptr, *attb = nullptr, *hbb = nullptr;  // bf16 mode
};

static inline int lin(const DevConv& c, const float* A, int lda, int rows, float* C, int ldc, int act, float slope,
                      hipStream_t s, const double* pe = nullptr, int period = 0) {
  return run_conv(c, CONV_K1, A, lda, 1, rows, C, ldc, act, slope, nullptr, 0, s, nullptr, -1, pe, period);
}

// bf16-operand Linear: A bf16 [rows, lda] -> C fp32 and/or Cb bf16.  N >= 256 goes to the large-tile
// LDS-DMA kernel (gemm_bf16_big.hip), narrow heads to the 4-wave tap-GEMM.
static int lin_b(const DevConv& c, const bf16_t* A, int lda, int rows, float* C, int ldc, bf16_t* Cb, int ldcb, int act,
                 float slope, hipStream_t s, const double* pe = nullptr, int period = 0) {
  PT_CHECK(c.wb != nullptr, "bf16 weights were not packed for this layer");
  PT_CHECK(lda >= c.Kb, "lin_b: input leading dim %d < packed K %d", lda, c.Kb);
  const bool small = c.N != 512 && rows <= option(OPT_SMALL_ROWS);     // few 256 x 256 tiles: the 4-wave kernel's smaller tiles fill more CUs
  // latency regime: the small-batch kernel takes any plain Linear (launch_gemm_big routes it there by the row count)
  const bool sm = rows <= option(OPT_SM_ROWS) && c.Kb % 128 == 0 && !pe && c.N % 8 == 0 && (act == ACT_NONE || slope == 0.f);
  if (sm || (!small && c.N >= 256 && c.N % 8 == 0 && (act == ACT_NONE || slope == 0.f))) {
    GemmBigArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = c.wb; a.Wt = c.wbt; a.bias = c.b; a.M = rows; a.N = c.N; a.K = c.Kb; a.Npad = c.Npad;
    a.epi = EPI_STD; a.C = C; a.ldc = ldc; a.Cb = Cb; a.ldcb = ldcb; a.pe = pe; a.pe_period = period;
    a.relu = act == ACT_LEAKY;
    a.flops = 2.0 * rows * c.N * c.Cin;
    return launch_gemm_big(a, s);
  }
  GemmArgsB a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.W = c.wb; a.bias = c.b; a.C = C; a.ldc = ldc; a.Cb = Cb; a.ldcb = ldcb;
  a.pe = pe; a.pe_period = period;
  a.M = rows; a.N = c.N; a.K = c.Kb; a.Npad = c.Npad;
  a.ntaps = 1; a.Lgrid = rows; a.Lin = rows; a.Lout = rows; a.in_stride = 1; a.out_stride = 1;
  a.act = act; a.slope = slope;
  a.zero_to = 0;
  a.flops = 2.0 * rows * c.N * c.Cin;
  return launch_gemm_bf16(a, s);
}

// bf16 mode, tail of every sub-layer: the sub-layer's last Linear `last` (A -> 512) with the style pre-op fused into
// its epilogue (tmp = SiLU(LN(.)(1+scale)+shift), bf16), then the style Linear with the post-LN residual fused
// (x = LN(x + .), fp32 + bf16 copy).  The pre-LayerNorm activations never reach HBM.
// bf16 mode, residual stream: x lives in bf16 only, so the ADDLN epilogue reads and writes 1 KB per row instead of
// reading 2 KB and writing 3 KB (116 -> 81 us per launch at 75,776 rows; the rounding of x once per sub-layer moves the
// logits' relative error from 5.9e-3 to 6.4e-3, tests/test_gpu_bf16.py).  Option resid_bf16 = 0 (pt_set_option) keeps the fp32
// residual stream with a bf16 shadow (A/B measurements).
static bool resid_bf16() { return option(OPT_RESID_BF16) == 1; }

//
// Where it pays, both launches become one (gemm_bf16_sub.hip: rows stay in registers between the two products).  That
// kernel runs one 128-row tile per CU per round of its persistent grid, ~50 us a round at K1 = 512 whatever the fill, and
// its epilogues are not hidden behind another wave's MFMAs, so it wins only when its last round is nearly full:
// measured (tools/sublayer_bench.py, profiles/r2/sublayer_vs_two_launches.txt) 250 vs 305 us at 153,600 rows (4.7
// rounds) but 143 vs 140 us at 75,776 (2.3 rounds) and 97 vs 74 us at 37,888 (1.2 rounds).  Rule: K1 = 512 and tiles
// >= 85 % of the rounds they occupy (90 % at K1 = 2048, where it gains 4-11 %).  Option fused_tail = 0 / 1
// (pt_set_option) forces it off / on (A/B measurements, tests).
static bool use_fused_tail(const DevConv& last, const StyleW& st, int rows, int n_classes) {
  if (!resid_bf16() || !st.out_wp || n_classes > 4 || last.Npad != DM || last.Kb % 32 != 0 || last.Kb < 64) return false;
  const int mode = option(OPT_FUSED_TAIL);
  if (mode >= 0) return mode == 1;
  const long long tiles = (rows + 127) / 128, grid = sublayer_grid_limit(), rounds = (tiles + grid - 1) / grid;
  return tiles * 100 >= rounds * grid * (last.Kb == DM ? 85 : 90);
}

static int sublayer_tail_b(const DevConv& last, const bf16_t* A, int lda, bf16_t* tmp, float* tf, const StyleW& st, const LNW& ln,
                           float* x, bf16_t* xb, int rows, int rows_per_clip, const int64_t* label, int n_label,
                           int n_classes, hipStream_t s) {
  PT_CHECK(last.wb && st.out.wb && last.N == DM && st.out.N == DM && st.out.Kb == DM, "sublayer_tail: unexpected geometry");
  {
    // Latency regime (1-16 clips): the two Linears on the small-batch kernel (64 x 64 tiles, 8 workgroups per 64 rows, the
    // operands of the whole K in flight at once) with fp32 outputs in `tf`, the two normalisations as the fp32 mode's row
    // kernels behind them: four short launches instead of two whose length is a 3-stage ring walking the whole W.
    GemmBigArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.W = last.wb; g.bias = last.b; g.M = rows; g.N = DM; g.K = last.Kb; g.Npad = last.Npad;
    g.epi = EPI_STD; g.C = tf; g.ldc = DM; g.flops = 2.0 * rows * DM * last.Cin;
    if (tf && gemm_sm_wants(g)) {
      if (launch_gemm_big(g, s)) return 1;
      if (launch_ln_mod_silu(tf, nullptr, tmp, st.ln.g, st.ln.b, st.table, label, n_label, rows, rows_per_clip, n_classes, s)) return 1;
      g.A = tmp; g.lda = DM; g.W = st.out.wb; g.bias = st.out.b; g.K = DM; g.Npad = st.out.Npad; g.flops = 2.0 * rows * DM * DM;
      if (launch_gemm_big(g, s)) return 1;
      if (resid_bf16()) return launch_add_ln_b(xb, tf, ln.g, ln.b, rows, s);
      return launch_add_ln(x, tf, ln.g, ln.b, rows, xb, s);
    }
  }
  if (use_fused_tail(last, st, rows, n_classes)) {
    SubArgs f;
    memset(&f, 0, sizeof(f));
    f.A = A; f.lda = lda; f.K1 = last.Kb; f.W1 = last.wb; f.b1 = last.b; f.fold = st.fold;
    f.label = label; f.n_label = n_label; f.rows_per_clip = rows_per_clip; f.n_classes = n_classes;
    f.W2p = st.out_wp; f.b2 = st.out.b; f.gamma = ln.g; f.beta = ln.b; f.X = xb; f.M = rows;
    return launch_sublayer(f, s);
  }
  GemmBigArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.W = last.wb; a.Wt = last.wbt; a.bias = last.b; a.M = rows; a.N = DM; a.K = last.Kb; a.Npad = last.Npad;
  a.epi = EPI_LNMOD; a.Cb = tmp; a.ldcb = DM; a.style = st.fold;
  a.label = label; a.n_label = n_label; a.rows_per_clip = rows_per_clip; a.n_classes = n_classes;
  a.flops = 2.0 * rows * DM * last.Cin;
  if (launch_gemm_big(a, s)) return 1;
  memset(&a, 0, sizeof(a));
  a.A = tmp; a.lda = DM; a.W = st.out.wb; a.Wt = st.out.wbt; a.bias = st.out.b; a.M = rows; a.N = DM; a.K = DM; a.Npad = st.out.Npad;
  a.epi = EPI_ADDLN; a.X = resid_bf16() ? nullptr : x; a.Cb = xb; a.ldcb = DM; a.gamma = ln.g; a.beta = ln.b;
  a.flops = 2.0 * rows * DM * DM;
  return launch_gemm_big(a, s);
}

static int style_then_ln(const StyleW& st, const LNW& ln, TBufs& b, int rows, int rows_per_clip, const int64_t* label,
                         int n_label, int n_classes, hipStream_t s) {
  // fp32 mode: b.t holds the sub-layer output; x <- LN(x + style(t))
  if (launch_ln_mod_silu(b.t, b.att, nullptr, st.ln.g, st.ln.b, st.table, label, n_label, rows, rows_per_clip,
                         n_classes, s)) return 1;
  if (lin(st.out, b.att, DM, rows, b.t, DM, ACT_NONE, 1.f, s)) return 1;
  return launch_add_ln(b.x, b.t, ln.g, ln.b, rows, nullptr, s);
}

// kv: cross-attention K | V rows [Rm, 1024]: floats in fp32 mode, bf16 in bf16 mode
static int decoder_layer(const DecLayer& L, TBufs& b, const void* kv, int B, int Nq, int Nk,
                         const int64_t* label, int n_label, int n_classes, bool bf, hipStream_t s) {
  const int R = B * Nq;
  if (!bf) {
    const float* kvf = (const float*)kv;
    if (lin(L.sa_in, b.x, DM, R, b.qkv, 3 * DM, ACT_NONE, 1.f, s)) return 1;
    if (launch_attention(b.qkv, 3 * DM, b.qkv + DM, 3 * DM, b.qkv + 2 * DM, 3 * DM, b.att, DM, B, Nq, Nq, NH, s)) return 1;
    if (lin(L.sa_out, b.att, DM, R, b.t, DM, ACT_NONE, 1.f, s)) return 1;
    if (style_then_ln(L.st[0], L.n1, b, R, Nq, label, n_label, n_classes, s)) return 1;
    if (lin(L.ca_q, b.x, DM, R, b.qkv, DM, ACT_NONE, 1.f, s)) return 1;
    if (launch_attention(b.qkv, DM, kvf, 2 * DM, kvf + DM, 2 * DM, b.att, DM, B, Nq, Nk, NH, s)) return 1;
    if (lin(L.ca_out, b.att, DM, R, b.t, DM, ACT_NONE, 1.f, s)) return 1;
    if (style_then_ln(L.st[1], L.n2, b, R, Nq, label, n_label, n_classes, s)) return 1;
    if (lin(L.ff1, b.x, DM, R, b.hb, FF, ACT_LEAKY, 0.f, s)) return 1;
    if (lin(L.ff2, b.hb, FF, R, b.t, DM, ACT_NONE, 1.f, s)) return 1;
    return style_then_ln(L.st[2], L.n3, b, R, Nq, label, n_label, n_classes, s);
  }
  const bf16_t* kvb = (const bf16_t*)kv;
  if (lin_b(L.sa_in, b.xb, DM, R, nullptr, 0, b.qkvb, 3 * DM, ACT_NONE, 1.f, s)) return 1;
  if (launch_attention_bf16(b.qkvb, 3 * DM, b.qkvb + DM, 3 * DM, b.qkvb + 2 * DM, 3 * DM, b.attb, DM, B, Nq, Nq, NH, s)) return 1;
  if (sublayer_tail_b(L.sa_out, b.attb, DM, b.hbb, b.t, L.st[0], L.n1, b.x, b.xb, R, Nq, label, n_label, n_classes, s)) return 1;
  if (lin_b(L.ca_q, b.xb, DM, R, nullptr, 0, b.qkvb, DM, ACT_NONE, 1.f, s)) return 1;
  if (launch_attention_bf16(b.qkvb, DM, kvb, 2 * DM, kvb + DM, 2 * DM, b.attb, DM, B, Nq, Nk, NH, s)) return 1;
  if (sublayer_tail_b(L.ca_out, b.attb, DM, b.hbb, b.t, L.st[1], L.n2, b.x, b.xb, R, Nq, label, n_label, n_classes, s)) return 1;
  if (lin_b(L.ff1, b.xb, DM, R, nullptr, 0, b.hbb, FF, ACT_LEAKY, 0.f, s)) return 1;
  return sublayer_tail_b(L.ff2, b.hbb, FF, b.attb, b.t, L.st[2], L.n3, b.x, b.xb, R, Nq, label, n_label, n_classes, s);
}

static size_t vt_elems(int B, int Nk) { return (size_t)B * NH * 64 * round_up(Nk, 32); }

static TBufs alloc_tbufs(Arena& ar, size_t R, int B, int Nq, bool bf) {
  TBufs b;
  b.x = ar.f32(R * DM);
  b.t = ar.f32(R * DM);
  if (!bf) {
    b.qkv = ar.f32(R * 3 * DM);
    b.att = ar.f32(R * DM);
    b.hb = ar.f32(R * FF);
  } else {
    b.xb = (bf16_t*)ar.raw(R * DM * 2);
    b.qkvb = (bf16_t*)ar.raw(R * 3 * DM * 2);
    b.attb = (bf16_t*)ar.raw(R * DM * 2);
    b.hbb = (bf16_t*)ar.raw(R * FF * 2);
  }
  return b;
}

// ------------------------------------------------------------------ predictor
static int tq_of(int T) {
  for (int i = 0; i < 3; ++i) T = (T - 2) / 2 + 1;
  return T;
}

size_t predictor_workspace_bytes(const Model& m, int B, int T) {
  if (!m.pred) return 0;
  const Predictor& P = *m.pred;
  const int Tq = (T + 7) / 8;
  const size_t R = (size_t)B * Tq * P.groups;
  const size_t cond_phase = (size_t)B * T * (ld_of(P.mot_dim + 1) + 4 * 128);
  // fp32 sizing covers the bf16 mode too (its buffers are half as wide, plus the small V^T scratch)
  const size_t trans = (size_t)B * Tq * (2 * (size_t)P.cond_w + DM + (size_t)P.n_layers * 3 * DM) +
                       R * (DM * 7 + FF + P.num_codes + 1) + vt_elems(B, Tq * P.groups);
  return (std::max(cond_phase + (size_t)B * Tq * P.cond_w, trans) * sizeof(float) + 96 * 256 + 255) & ~(size_t)255;
}

// motion-context encoder; writes channels-last condition rows [B*Tq, cond_w] (motion | audio [| text])
static int predictor_condition_cl(const Predictor& P, const float* masked_motion, const float* mask, const float* audio,
                                  int B, int T, float* cond_cl, Arena& ar, hipStream_t s) {
  const int ldm = ld_of(P.mot_dim + 1);
  const size_t stage = (size_t)B * T * 128;
  float* mm = ar.f32((size_t)B * T * ldm);
  float *h = ar.f32(stage), *g = ar.f32(stage), *sk = ar.f32(stage), *o = ar.f32(stage);
  PT_CHECK(!ar.overflow, "predictor_condition: workspace too small (%zu given, %zu needed)", ar.cap, ar.off);
  if (launch_ncl_to_cl(masked_motion, mm, B, P.mot_dim, T, ldm, 0, P.mot_dim, s)) return 1;
  if (launch_ncl_to_cl(mask, mm, B, 1, T, ldm, P.mot_dim, ldm, s)) return 1;
  if (lin(P.motion_emb, mm, ldm, B * T, h, ld_of(P.motion_emb.N), ACT_NONE, 1.f, s)) return 1;
  int L = T;
  for (int i = 0; i < 10; ++i) {
    const GatedW& gw = P.enc[i];
    const int ldi = ld_of(gw.in_dim), ldd = ld_of(gw.dim);
    int Lo = L;
    if (run_conv(gw.vert, gw.kind, h, ldi, B, L, g, ldd, ACT_GATED, 1.f, nullptr, 0, s, &Lo)) return 1;
    const float* res = h;
    int ldr = ldi;
    if (gw.strided) {
      if (run_conv(gw.skip, CONV_DOWN, h, ldi, B, L, sk, ldd, ACT_NONE, 1.f, nullptr, 0, s)) return 1;
      res = sk;
      ldr = ldd;
    }
    const bool last = i == 9;
    float* dst = last ? cond_cl : o;
    if (run_conv(gw.res, CONV_K1, g, ldd, B, Lo, dst, last ? P.cond_w : ldd, ACT_NONE, 1.f, res, ldr, s, nullptr,
                 last ? gw.dim : -1)) return 1;
    if (!last) { float* t = h; h = o; o = t; }
    L = Lo;
  }
  // audio (+ text) features (B, audio_dim, Tq) NCL -> columns [512, cond_w)
  return launch_ncl_to_cl(audio, cond_cl, B, P.audio_dim, L, P.cond_w, P.enc[9].dim, P.enc[9].dim + P.audio_dim, s);
}

// bf16 mode of the motion-context encoder: bf16 rows between the gated convs, fp32 condition rows at the end
static int predictor_condition_cl_b(const Predictor& P, const float* masked_motion, const float* mask, const float* audio,
                                    int B, int T, float* cond_cl, Arena& ar, hipStream_t s) {
  const int ldm = round_up(P.mot_dim + 1, 64);
  const size_t stage = (size_t)B * T * 128;
  bf16_t* mm = (bf16_t*)ar.raw((size_t)B * T * ldm * 2);
  bf16_t *h = (bf16_t*)ar.raw(stage * 2), *g = (bf16_t*)ar.raw(stage * 2), *sk = (bf16_t*)ar.raw(stage * 2),
         *o = (bf16_t*)ar.raw(stage * 2);
  PT_CHECK(!ar.overflow, "predictor_condition: workspace too small (%zu given, %zu needed)", ar.cap, ar.off);
  if (launch_ncl_to_cl_b(masked_motion, mm, B, P.mot_dim, T, ldm, 0, P.mot_dim, s)) return 1;
  if (launch_ncl_to_cl_b(mask, mm, B, 1, T, ldm, P.mot_dim, ldm, s)) return 1;
  PT_CHECK(P.motion_emb.N % 64 == 0, "predictor_condition(bf16): motion embedding width must be a multiple of 64");
  if (run_conv_b(P.motion_emb, CONV_K1, mm, ldm, B, T, nullptr, 0, h, P.motion_emb.N, ACT_NONE, 1.f, nullptr, 0, s)) return 1;
  int L = T;
  for (int i = 0; i < 10; ++i) {
    const GatedW& gw = P.enc[i];
    const int ldi = gw.in_dim, ldd = gw.dim;
    PT_CHECK(ldi % 64 == 0 && ldd % 64 == 0, "predictor_condition(bf16): gated conv widths must be multiples of 64");
    int Lo = L;
    if (run_conv_b(gw.vert, gw.kind, h, ldi, B, L, nullptr, 0, g, ldd, ACT_GATED, 1.f, nullptr, 0, s, &Lo)) return 1;
    const bf16_t* res = h;
    int ldr = ldi;
    if (gw.strided) {
      if (run_conv_b(gw.skip, CONV_DOWN, h, ldi, B, L, nullptr, 0, sk, ldd, ACT_NONE, 1.f, nullptr, 0, s)) return 1;
      res = sk;
      ldr = ldd;
    }
    const bool last = i == 9;
    if (run_conv_b(gw.res, CONV_K1, g, ldd, B, Lo, last ? cond_cl : nullptr, last ? P.cond_w : 0, last ? nullptr : o, ldd,
                   ACT_NONE, 1.f, res, ldr, s, nullptr, gw.dim)) return 1;
    if (!last) { bf16_t* t = h; h = o; o = t; }
    L = Lo;
  }
  return launch_ncl_to_cl(audio, cond_cl, B, P.audio_dim, L, P.cond_w, P.enc[9].dim, P.enc[9].dim + P.audio_dim, s);
}

int predictor_condition(Model& m, const float* masked_motion, const float* mask, const float* audio, int B, int T,
                        float* condition, void* ws, size_t ws_bytes, hipStream_t s) {
  PT_CHECK(m.finalized && m.pred, "predictor_condition: wrong or unfinalized model");
  PT_CHECK(B > 0 && T >= 8, "predictor_condition: need B>0, T>=8 (B=%d T=%d)", B, T);
  const Predictor& P = *m.pred;
  Arena ar(ws, ws_bytes);
  const int Tq = tq_of(T);
  float* cond_cl = ar.f32((size_t)B * Tq * P.cond_w);
  if (m.precision == 1 ? predictor_condition_cl_b(P, masked_motion, mask, audio, B, T, cond_cl, ar, s)
                       : predictor_condition_cl(P, masked_motion, mask, audio, B, T, cond_cl, ar, s)) return 1;
  return launch_cl_to_ncl(cond_cl, P.cond_w, 0, condition, B, P.cond_w, Tq, s);
}

struct PredMem {
  float* mem = nullptr;                 // fp32 mode
  bf16_t *condb = nullptr, *memb = nullptr;   // bf16 mode
  std::vector<void*> kv;
};

static int alloc_predmem(const Predictor& P, Arena& ar, int B, int Tq, bool bf, PredMem& pm) {
  const size_t Rm = (size_t)B * Tq;
  pm.kv.resize(P.n_layers);
  if (!bf) {
    pm.mem = ar.f32(Rm * DM);
    for (int l = 0; l < P.n_layers; ++l) pm.kv[l] = ar.f32(Rm * 2 * DM);
  } else {
    pm.condb = (bf16_t*)ar.raw(Rm * P.cond_w * 2);
    pm.memb = (bf16_t*)ar.raw(Rm * DM * 2);
    for (int l = 0; l < P.n_layers; ++l) {
      pm.kv[l] = ar.raw(Rm * 2 * DM * 2);
    }
  }
  return 0;
}

// condition rows -> memory (condi_emb + fp64 PE) and per-layer cross-attention K|V (+ V^T in bf16 mode)
static int predictor_memory(const Predictor& P, const float* cond_cl, int B, int Tq, bool bf, PredMem& pm,
                            hipStream_t s) {
  PT_CHECK(Tq + 1 <= P.pe_rows, "sequence of %d latent steps exceeds the positional table (%d)", Tq, P.pe_rows - 1);
  const int Rm = B * Tq;
  if (!bf) {
    if (lin(P.condi_emb, cond_cl, P.cond_w, Rm, pm.mem, DM, ACT_NONE, 1.f, s, P.pe + DM, Tq)) return 1;
    for (int l = 0; l < P.n_layers; ++l)
      if (lin(P.layers[l].ca_kv, pm.mem, DM, Rm, (float*)pm.kv[l], 2 * DM, ACT_NONE, 1.f, s)) return 1;
    return 0;
  }
  if (launch_cvt_f32_bf16(cond_cl, P.cond_w, pm.condb, P.cond_w, Rm, P.cond_w, P.cond_w, s)) return 1;
  if (lin_b(P.condi_emb, pm.condb, P.cond_w, Rm, nullptr, 0, pm.memb, DM, ACT_NONE, 1.f, s, P.pe + DM, Tq)) return 1;
  for (int l = 0; l < P.n_layers; ++l) {
    bf16_t* kvb = (bf16_t*)pm.kv[l];
    if (lin_b(P.layers[l].ca_kv, pm.memb, DM, Rm, nullptr, 0, kvb, 2 * DM, ACT_NONE, 1.f, s)) return 1;
  }
  return 0;
}

static int predictor_pass(const Predictor& P, const int* ids, const int64_t* label, const PredMem& pm, TBufs& tb, int B,
                          int Tq, bool bf, float* logits, hipStream_t s) {
  const int N = Tq * P.groups, R = B * N;
  if (launch_embed_pe(ids, P.tok_emb, P.pe, P.ppe, bf && resid_bf16() ? nullptr : tb.x, tb.xb, R, N, P.groups, P.num_codes + 1, s)) return 1;
  for (int l = 0; l < P.n_layers; ++l)
    if (decoder_layer(P.layers[l], tb, pm.kv[l], B, N, Tq, label, B, P.n_classes, bf, s)) return 1;
  if (!bf) {
    if (lin(P.out0, tb.x, DM, R, tb.t, DM, ACT_LEAKY, 0.f, s)) return 1;
    return lin(P.out2, tb.t, DM, R, logits, P.num_codes, ACT_NONE, 1.f, s);
  }
  if (lin_b(P.out0, tb.xb, DM, R, nullptr, 0, tb.attb, DM, ACT_LEAKY, 0.f, s)) return 1;
  return lin_b(P.out2, tb.attb, DM, R, logits, P.num_codes, nullptr, 0, ACT_NONE, 1.f, s);
}

int predictor_logits(Model& m, const int32_t* ids, const int64_t* label, const float* condition, int B, int Tq,
                     float* logits, void* ws, size_t ws_bytes, hipStream_t s) {
  PT_CHECK(m.finalized && m.pred, "predictor_logits: wrong or unfinalized model");
  PT_CHECK(B > 0 && Tq > 0, "predictor_logits: empty input B=%d T'=%d", B, Tq);
  const Predictor& P = *m.pred;
  const bool bf = m.precision == 1;
  Arena ar(ws, ws_bytes);
  float* cond_cl = ar.f32((size_t)B * Tq * P.cond_w);
  PredMem pm;
  alloc_predmem(P, ar, B, Tq, bf, pm);
  TBufs tb = alloc_tbufs(ar, (size_t)B * Tq * P.groups, B, Tq * P.groups, bf);
  PT_CHECK(!ar.overflow, "predictor_logits: workspace too small (%zu given, %zu needed)", ws_bytes, ar.off);
  if (launch_ncl_to_cl(condition, cond_cl, B, P.cond_w, Tq, P.cond_w, 0, P.cond_w, s)) return 1;
  if (predictor_memory(P, cond_cl, B, Tq, bf, pm, s)) return 1;
  return predictor_pass(P, ids, label, pm, tb, B, Tq, bf, logits, s);
}

// noise_q / noise_g NULL: the MaskGIT step kernel generates both draws itself from (seed, step, element); seed_dev, when
// given, is where the kernels read the seed from at run time (graph replay with a fresh seed per replay)
int predictor_sample(Model& m, const int64_t* label, const float* condition, const float* noise_q, const float* noise_g,
                     int steps, int B, int Tq, int32_t* codes_out, float* logits_out, void* ws, size_t ws_bytes,
                     hipStream_t s, unsigned long long seed, const unsigned long long* seed_dev) {
  PT_CHECK(m.finalized && m.pred, "predictor_sample: wrong or unfinalized model");
  PT_CHECK(B > 0 && Tq > 0 && steps > 0, "predictor_sample: empty input B=%d T'=%d steps=%d", B, Tq, steps);
  const Predictor& P = *m.pred;
  const bool bf = m.precision == 1;
  const int N = Tq * P.groups, V = P.num_codes;
  const size_t R = (size_t)B * N;
  Arena ar(ws, ws_bytes);
  float* cond_cl = ar.f32((size_t)B * Tq * P.cond_w);
  PredMem pm;
  alloc_predmem(P, ar, B, Tq, bf, pm);
  TBufs tb = alloc_tbufs(ar, R, B, N, bf);
  int* ids = (int*)ar.raw(R * sizeof(int));
  float* lg = logits_out ? nullptr : ar.f32(R * V);
  PT_CHECK(!ar.overflow, "predictor_sample: workspace too small (%zu given, %zu needed)", ws_bytes, ar.off);
  if (launch_ncl_to_cl(condition, cond_cl, B, P.cond_w, Tq, P.cond_w, 0, P.cond_w, s)) return 1;
  if (predictor_memory(P, cond_cl, B, Tq, bf, pm, s)) return 1;
  if (launch_fill_i32(ids, V, R, s)) return 1;
  for (int t = 0; t < steps; ++t) {
    float* lt = logits_out ? logits_out + (size_t)t * R * V : lg;
    if (predictor_pass(P, ids, label, pm, tb, B, Tq, bf, lt, s)) return 1;
    // schedule of PTransformer.py:613-636: cosine mask ratio, fp32 product floor, annealed Gumbel temperature
    const double ratio = 1.0 * (t + 1) / steps;
    const float mask_ratio = (float)cos(ratio * M_PI / 2);
    const int want = (int)floorf((float)N * mask_ratio);
    const float temp = (float)(4.5 * (1.0 - ratio));
    if (launch_maskgit_step(lt, noise_q ? noise_q + (size_t)t * R * V : nullptr, noise_g ? noise_g + (size_t)t * R : nullptr, ids, B,
                            N, V, V, want, temp, s, seed, t, seed_dev)) return 1;
  }
  PT_HIP(hipMemcpyAsync(codes_out, ids, R * sizeof(int), hipMemcpyDeviceToDevice, s));
  return 0;
}

// ------------------------------------------------------------------ refiner
size_t refiner_workspace_bytes(const Model& m, int B, int T) {
  if (!m.ref) return 0;
  const Refiner& R = *m.ref;
  const size_t rows = (size_t)B * T;
  const size_t f = rows * (ld_of(R.in_dim) * 3 + 2 * round_up(R.cond_dim + 2, 64) + DM + 2 * DM + 7 * DM + FF) +
                   2 * vt_elems(B, T);
  return (f * sizeof(float) + 64 * 256 + 255) & ~(size_t)255;
}

int refiner_forward(Model& m, const float* x, const float* condition, const float* mask, const int64_t* label,
                    int n_label, int B, int T, float* out, void* ws, size_t ws_bytes, hipStream_t s) {
  PT_CHECK(m.finalized && m.ref, "refiner_forward: wrong or unfinalized model");
  PT_CHECK(B > 0 && T > 0, "refiner_forward: empty input B=%d T=%d", B, T);
  PT_CHECK(n_label == 1 || n_label == B, "refiner_forward: label must have 1 or B=%d entries (got %d)", B, n_label);
  const Refiner& R = *m.ref;
  const bool bf = m.precision == 1;
  PT_CHECK(T + 1 <= R.pe_rows, "sequence of %d frames exceeds the positional table (%d)", T, R.pe_rows - 1);
  const int rows = B * T, ldx = ld_of(R.in_dim), ldc = round_up(R.cond_dim + 2, 64);
  Arena ar(ws, ws_bytes);
  float* x_cl = ar.f32((size_t)rows * ldx);
  float* c_cl = ar.f32((size_t)rows * ldc);
  float* o_cl = ar.f32((size_t)rows * ldx);
  TBufs tb = alloc_tbufs(ar, rows, B, T, bf);
  float *mem = nullptr, *kv = nullptr;
  bf16_t *xb_cl = nullptr, *cb_cl = nullptr, *memb = nullptr, *kvb = nullptr, *h1b = nullptr;
  if (!bf) {
    mem = ar.f32((size_t)rows * DM);
    kv = ar.f32((size_t)rows * 2 * DM);
  } else {
    xb_cl = (bf16_t*)ar.raw((size_t)rows * ldx * 2);
    cb_cl = (bf16_t*)ar.raw((size_t)rows * ldc * 2);
    memb = (bf16_t*)ar.raw((size_t)rows * DM * 2);
    kvb = (bf16_t*)ar.raw((size_t)rows * 2 * DM * 2);
    h1b = (bf16_t*)ar.raw((size_t)rows * 1024 * 2);
  }
  PT_CHECK(!ar.overflow, "refiner_forward: workspace too small (%zu given, %zu needed)", ws_bytes, ar.off);
  if (launch_ncl_to_cl(x, x_cl, B, R.in_dim, T, ldx, 0, ldx, s)) return 1;
  if (launch_ncl_to_cl(condition, c_cl, B, R.cond_dim, T, ldc, 0, R.cond_dim, s)) return 1;
  if (launch_mask_onehot(mask, c_cl, rows, ldc, R.cond_dim, ldc, s)) return 1;
  if (!bf) {
    if (lin(R.input_conv, x_cl, ldx, rows, tb.x, DM, ACT_NONE, 1.f, s, R.pe + DM, T)) return 1;
    if (lin(R.audio_conv, c_cl, ldc, rows, mem, DM, ACT_NONE, 1.f, s, R.pe + DM, T)) return 1;
    for (int l = 0; l < R.n_layers; ++l) {
      if (lin(R.layers[l].ca_kv, mem, DM, rows, kv, 2 * DM, ACT_NONE, 1.f, s)) return 1;
      if (decoder_layer(R.layers[l], tb, kv, B, T, T, label, n_label, R.n_classes, false, s)) return 1;
    }
    if (lin(R.out0, tb.x, DM, rows, tb.hb, 1024, ACT_LEAKY, 0.f, s)) return 1;
    if (lin(R.out2, tb.hb, 1024, rows, o_cl, ldx, ACT_NONE, 1.f, s)) return 1;
  } else {
    if (launch_cvt_f32_bf16(x_cl, ldx, xb_cl, ldx, rows, ldx, ldx, s)) return 1;
    if (launch_cvt_f32_bf16(c_cl, ldc, cb_cl, ldc, rows, ldc, ldc, s)) return 1;
    if (lin_b(R.input_conv, xb_cl, ldx, rows, resid_bf16() ? nullptr : tb.x, DM, tb.xb, DM, ACT_NONE, 1.f, s, R.pe + DM, T)) return 1;
    if (lin_b(R.audio_conv, cb_cl, ldc, rows, nullptr, 0, memb, DM, ACT_NONE, 1.f, s, R.pe + DM, T)) return 1;
    for (int l = 0; l < R.n_layers; ++l) {
      if (lin_b(R.layers[l].ca_kv, memb, DM, rows, nullptr, 0, kvb, 2 * DM, ACT_NONE, 1.f, s)) return 1;
      if (decoder_layer(R.layers[l], tb, kvb, B, T, T, label, n_label, R.n_classes, true, s)) return 1;
    }
    if (lin_b(R.out0, tb.xb, DM, rows, nullptr, 0, h1b, 1024, ACT_LEAKY, 0.f, s)) return 1;
    if (lin_b(R.out2, h1b, 1024, rows, o_cl, ldx, nullptr, 0, ACT_NONE, 1.f, s)) return 1;
    // o_cl pad columns [in_dim, ldx) stay unwritten: cl_to_ncl reads only the first in_dim columns
  }
  return launch_cl_to_ncl(o_cl, ldx, 0, out, B, R.in_dim, T, s);
}

}  // namespace pt
