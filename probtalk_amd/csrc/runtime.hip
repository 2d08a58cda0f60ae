// Weight folding/packing and the conv-stack module sequences (PQ-VAE, legacy VQ-VAE, audio
// condition encoder).  Reference structure followed: nets/inpainting/vqvae_1d_sc.py:70-91,197-324,
// nets/spg/vqvae_1d.py:66-187, nets/spg/vqvae_modules.py:107-245 (eval mode).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "model.h"

namespace pt {

// ------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

void free_transformer_models(Model& m);  // transformer.hip

Model::~Model() {
  free_transformer_models(*this);
  for (void* p : allocs) (void)hipFree(p);
}

// ------------------------------------------------------------------ packing
static const HostTensor* find(const Model& m, const std::string& name) {
  auto it = m.sd.find(name);
  return it == m.sd.end() ? nullptr : &it->second;
}

template <typename T>
static int upload(Model& m, const std::vector<T>& h, T** dev) {
  void* p = nullptr;
  PT_HIP(hipMalloc(&p, h.size() * sizeof(T)));
  m.allocs.push_back(p);
  PT_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *dev = (T*)p;
  return 0;
}

// Pack conv/linear weights as [slab][Npad][K] (K = round_up(Cin,32), zero padded), folding an
// eval-mode BatchNorm (w*s, (b-mean)*s+beta) and optionally adding a second conv of identical
// geometry applied to the same input (the 'residual_layer' of down/up ConvNormRelu blocks,
// vqvae_modules.py:142-156,196-198): BN(conv1(x)) + conv2(x) == conv_{W1*s+W2}(x) + b'.
static int pack_conv(Model& m, DevConv& out, const std::string& w1, const std::string& bn, const std::string& w2,
                     bool transposed, bool gated) {
  const HostTensor* W = find(m, w1 + ".weight");
  PT_CHECK(W != nullptr, "missing tensor %s.weight", w1.c_str());
  PT_CHECK(W->shape.size() == 2 || W->shape.size() == 3, "%s.weight: expected 2-D or 3-D", w1.c_str());
  const int kt = W->shape.size() == 3 ? (int)W->shape[2] : 1;
  const int d0 = (int)W->shape[0], d1 = (int)W->shape[1];
  const int N = transposed ? d1 : d0, Cin = transposed ? d0 : d1;
  const HostTensor* Bv = find(m, w1 + ".bias");
  const HostTensor* W2 = w2.empty() ? nullptr : find(m, w2 + ".weight");
  const HostTensor* B2 = w2.empty() ? nullptr : find(m, w2 + ".bias");
  if (!w2.empty()) {
    PT_CHECK(W2 != nullptr, "missing tensor %s.weight", w2.c_str());
    PT_CHECK(W2->shape == W->shape, "%s.weight / %s.weight shape mismatch", w1.c_str(), w2.c_str());
  }
  std::vector<double> sc(N, 1.0), sh(N, 0.0);
  if (!bn.empty()) {
    const HostTensor *g = find(m, bn + ".weight"), *b = find(m, bn + ".bias"), *mu = find(m, bn + ".running_mean"),
                     *var = find(m, bn + ".running_var");
    PT_CHECK(g && b && mu && var, "missing BatchNorm tensors under %s", bn.c_str());
    PT_CHECK((int)g->numel() == N, "%s: BatchNorm width %d != conv out %d", bn.c_str(), (int)g->numel(), N);
    for (int n = 0; n < N; ++n) {
      // same fp32 inverse-std the eval BatchNorm uses, then exact products in double
      const float inv = 1.0f / sqrtf(var->f[n] + 1e-5f);
      sc[n] = (double)g->f[n] * (double)inv;
      sh[n] = (double)b->f[n] - (double)mu->f[n] * sc[n];
    }
  }
  out.N = N;
  out.Cin = Cin;
  out.K = round_up(Cin, 32);
  out.Npad = round_up(N, 128);
  out.slabs = kt;
  out.merged = W2 ? 2 : 1;
  std::vector<float> hw((size_t)kt * out.Npad * out.K, 0.f), hb(out.Npad, 0.f);
  const int dim = N / 2;
  auto prow = [&](int n) {
    if (!gated) return n;
    const int c = n < dim ? n : n - dim;
    return (c / 32) * 64 + (n < dim ? 0 : 32) + (c % 32);
  };
  if (gated) PT_CHECK(N % 64 == 0, "%s: gated conv needs 2*dim %% 64 == 0 (N=%d)", w1.c_str(), N);
  for (int n = 0; n < N; ++n) {
    const int pr = prow(n);
    for (int c = 0; c < Cin; ++c)
      for (int k = 0; k < kt; ++k) {
        const size_t src = transposed ? ((size_t)c * N + n) * kt + k : ((size_t)n * Cin + c) * kt + k;
        double v = (double)W->f[src] * sc[n];
        if (W2) v += (double)W2->f[src];
        hw[((size_t)k * out.Npad + pr) * out.K + c] = (float)v;
      }
    double bv = (Bv ? (double)Bv->f[n] : 0.0) * sc[n] + sh[n];
    if (B2) bv += (double)B2->f[n];
    hb[pr] = (float)bv;
  }
  if (upload(m, hw, &out.w)) return 1;
  if (m.precision == 1) {
    // bf16 copy (round to nearest even), [slab][Npad][Kb] with K padded to the bf16 kernels' 64-element stage
    out.Kb = round_up(Cin, 64);
    std::vector<unsigned short> hwb((size_t)kt * out.Npad * out.Kb, 0);
    for (int k = 0; k < kt; ++k)
      for (int n = 0; n < out.Npad; ++n)
        for (int c = 0; c < Cin; ++c) {
          uint32_t u;
          const float f = hw[((size_t)k * out.Npad + n) * out.K + c];
          memcpy(&u, &f, 4);
          u += 0x7fffu + ((u >> 16) & 1u);
          hwb[((size_t)k * out.Npad + n) * out.Kb + c] = (unsigned short)(u >> 16);
        }
    if (upload(m, hwb, &out.wb)) return 1;
    if (kt == 1 && out.N == 512 && out.Npad == 512) {
      // K-tile-major copy for the row kernel: its staging ring takes W 32 columns of k at a time, i.e. 64 bytes of every
      // row; in this order a DMA instruction's 16 rows x 64 B are 1 KB of consecutive memory (whole cache lines)
      std::vector<unsigned short> hwt((size_t)out.Npad * out.Kb);
      for (int t = 0; t < out.Kb / 32; ++t)
        for (int n = 0; n < out.Npad; ++n)
          memcpy(&hwt[((size_t)t * out.Npad + n) * 32], &hwb[(size_t)n * out.Kb + t * 32], 64);
      if (upload(m, hwt, &out.wbt)) return 1;
    }
  }
  return upload(m, hb, &out.b);
}

int pack_conv_public(Model& m, DevConv& out, const std::string& w1, const std::string& bn, const std::string& w2,
                     bool transposed, bool gated) {
  return pack_conv(m, out, w1, bn, w2, transposed, gated);
}

static int pack_stack(Model& m, ResStack& st, const std::string& p) {
  if (pack_conv(m, st.l0, p + "_layers.0.conv", p + "_layers.0.norm", "", false, false)) return 1;
  if (pack_conv(m, st.l1, p + "_layers.1.conv", p + "_layers.1.norm", "", false, false)) return 1;
  return pack_conv(m, st.fin, p + "conv", p + "norm", "", false, false);
}

static int finalize_vqvae(Model& m) {
  VQVAE& v = m.vq;
  PT_CHECK(m.cfg.size() >= 6, "vqvae: cfg = {in_dim, emb_dim, num_codes, num_hiddens, groups, levels}");
  v.in_dim = m.cfg[0]; v.emb_dim = m.cfg[1]; v.num_codes = m.cfg[2];
  const int nh = m.cfg[3];
  v.groups = m.cfg[4]; v.levels = m.cfg[5];
  const bool legacy = (m.kind == "vqvae_legacy");
  PT_CHECK(v.levels == (legacy ? 3 : 4), "vqvae: levels=%d unsupported for kind %s", v.levels, m.kind.c_str());
  PT_CHECK(v.groups >= 1 && v.emb_dim % v.groups == 0 && (v.emb_dim / v.groups) % 8 == 0,
           "vqvae: emb_dim=%d must split into %d groups of a multiple of 8", v.emb_dim, v.groups);
  if (legacy) v.ch = {nh / 4, nh / 2, nh};
  else v.ch = {nh / 4, nh / 4, nh / 2, nh};
  const int L = v.levels;
  if (pack_conv(m, v.enc_project, "encoder.project.conv", "encoder.project.norm", "", false, false)) return 1;
  v.enc.resize(L); v.dec.resize(L); v.down.resize(L - 1); v.up.resize(L - 1);
  for (int i = 0; i < L; ++i) {
    if (pack_stack(m, v.enc[i], "encoder._enc_" + std::to_string(i + 1) + ".")) return 1;
    if (pack_stack(m, v.dec[i], "decoder._dec_" + std::to_string(i + 1) + ".")) return 1;
  }
  for (int i = 0; i < L - 1; ++i) {
    const std::string d = "encoder._down_" + std::to_string(i + 1), u = "decoder._up_" + std::to_string(i + 2);
    if (pack_conv(m, v.down[i], d + ".conv", d + ".norm", d + ".residual_layer", false, false)) return 1;
    if (pack_conv(m, v.up[i], u + ".conv", u + ".norm", u + ".residual_layer", true, false)) return 1;
  }
  if (pack_conv(m, v.pre_vq, "encoder.pre_vq_conv", "", "", false, false)) return 1;
  if (pack_conv(m, v.aft_vq, "decoder.aft_vq_conv", "", "", false, false)) return 1;
  if (pack_conv(m, v.dec_project, "decoder.project", "", "", false, false)) return 1;
  const int D = v.emb_dim / v.groups;
  std::vector<float> cb((size_t)v.groups * v.num_codes * D);
  for (int g = 0; g < v.groups; ++g) {
    const std::string name = legacy ? "vq_layer.embeddings" : "vq_layer.embeddings_" + std::to_string(g);
    const HostTensor* e = find(m, name);
    PT_CHECK(e != nullptr, "missing codebook %s", name.c_str());
    PT_CHECK(e->shape.size() == 2 && e->shape[0] == v.num_codes && e->shape[1] == D,
             "codebook %s: expected (%d,%d)", name.c_str(), v.num_codes, D);
    memcpy(cb.data() + (size_t)g * v.num_codes * D, e->f.data(), sizeof(float) * v.num_codes * D);
  }
  return upload(m, cb, &v.codebooks);
}

static int finalize_cond_encoder(Model& m) {
  CondEncoder& c = m.ce;
  PT_CHECK(m.cfg.size() >= 2, "cond_encoder: cfg = {in_dim, num_hiddens}");
  c.in_dim = m.cfg[0]; c.hid = m.cfg[1];
  if (pack_conv(m, c.feat, "feature_map.0", "feature_map.1", "", false, false)) return 1;
  for (int i = 0; i < 3; ++i) {
    const std::string d = "_down_" + std::to_string(i + 1);
    if (pack_conv(m, c.down[i], d + ".conv", d + ".norm", d + ".residual_layer", false, false)) return 1;
  }
  return 0;
}

int finalize_transformer_models(Model& m);  // transformer.hip

int finalize_model(Model& m) {
  PT_CHECK(!m.finalized, "model already finalized");
  int rc;
  if (m.kind == "vqvae_sc" || m.kind == "vqvae_legacy") rc = finalize_vqvae(m);
  else if (m.kind == "cond_encoder") rc = finalize_cond_encoder(m);
  else rc = finalize_transformer_models(m);
  if (rc) return rc;
  if (conv8_prepare()) return 1;
  PT_HIP(hipDeviceSynchronize());
  m.finalized = true;
  m.sd.clear();  // host copies are no longer needed
  return 0;
}

// ------------------------------------------------------------------ conv launcher
int run_conv(const DevConv& c, ConvKind kind, const float* A, int lda, int B, int Lin, float* C, int ldc, int act,
             float slope, const float* Res, int ldr, hipStream_t s, int* Lout_p, int zero_to, const double* pe,
             int pe_period) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.W = c.w; a.bias = c.b; a.Res = Res; a.ldr = ldr; a.C = C; a.ldc = ldc;
  a.pe = pe; a.pe_period = pe_period;
  a.N = c.N; a.K = c.K; a.Npad = c.Npad;
  a.act = act; a.slope = slope;
  a.Lin = Lin; a.in_stride = 1; a.out_stride = 1; a.out_off = 0;
  const int n_out = act == ACT_GATED ? c.N / 2 : c.N;
  a.zero_to = zero_to >= 0 ? zero_to : (ldc == ld_of(n_out) ? ldc : n_out);
  PT_CHECK(lda >= c.K, "conv: input leading dim %d < packed K %d", lda, c.K);
  int Lout = Lin;
  switch (kind) {
    case CONV_K1:
      PT_CHECK(c.slabs == 1, "conv: kernel size %d where 1 expected", c.slabs);
      a.ntaps = 1; a.tap_d[0] = 0; a.tap_w[0] = 0;
      break;
    case CONV_K3:
    case CONV_K7: {
      const int kt = kind == CONV_K3 ? 3 : 7;
      PT_CHECK(c.slabs == kt, "conv: kernel size %d where %d expected", c.slabs, kt);
      a.ntaps = kt;
      for (int k = 0; k < kt; ++k) { a.tap_d[k] = k - kt / 2; a.tap_w[k] = k; }
      break;
    }
    case CONV_DOWN:
      PT_CHECK(c.slabs == 4, "conv: kernel size %d where 4 expected", c.slabs);
      PT_CHECK(Lin >= 2, "conv(down): input length %d too short", Lin);
      a.ntaps = 4; a.in_stride = 2;
      for (int k = 0; k < 4; ++k) { a.tap_d[k] = k - 1; a.tap_w[k] = k; }
      Lout = (Lin - 2) / 2 + 1;
      break;
    case CONV_UP:
      PT_CHECK(c.slabs == 4, "conv: kernel size %d where 4 expected", c.slabs);
      Lout = 2 * Lin;
      break;
  }
  a.Lout = Lout;
  if (Lout_p) *Lout_p = Lout;
  if (kind != CONV_UP) {
    a.Lgrid = Lout; a.M = B * Lout;
    a.flops = 2.0 * a.M * c.N * c.Cin * a.ntaps * c.merged;
    return launch_gemm(a, s);
  }
  // ConvTranspose1d k4 s2 p1: out[2q] = x[q] w1 + x[q-1] w3 ; out[2q+1] = x[q] w2 + x[q+1] w0
  a.Lgrid = Lin; a.M = B * Lin; a.out_stride = 2; a.ntaps = 2;
  a.flops = 2.0 * a.M * c.N * c.Cin * 2 * c.merged;
  a.out_off = 0; a.tap_d[0] = 0; a.tap_w[0] = 1; a.tap_d[1] = -1; a.tap_w[1] = 3;
  if (launch_gemm(a, s)) return 1;
  a.out_off = 1; a.tap_d[0] = 0; a.tap_w[0] = 2; a.tap_d[1] = 1; a.tap_w[1] = 0;
  return launch_gemm(a, s);
}

// bf16-operand variant of run_conv (bf16 mode of the conv stacks): A, the residual and the main output are bf16
// channels-last rows with ld = round_up(C, 64); an fp32 copy of the output is optional (module boundaries).
int run_conv_b(const DevConv& c, ConvKind kind, const bf16_t* A, int lda, int B, int Lin, float* C, int ldc, bf16_t* Cb,
               int ldcb, int act, float slope, const bf16_t* ResB, int ldr, hipStream_t s, int* Lout_p, int zero_to) {
  PT_CHECK(c.wb != nullptr, "conv: bf16 weights were not packed (model precision is fp32)");
  GemmArgsB a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.W = c.wb; a.bias = c.b; a.ResB = ResB; a.ldr = ldr; a.C = C; a.ldc = ldc; a.Cb = Cb; a.ldcb = ldcb;
  a.N = c.N; a.K = c.Kb; a.Npad = c.Npad;
  a.act = act; a.slope = slope;
  a.Lin = Lin; a.in_stride = 1; a.out_stride = 1; a.out_off = 0;
  const int n_out = act == ACT_GATED ? c.N / 2 : c.N;
  a.zero_to = zero_to >= 0 ? zero_to : n_out;
  PT_CHECK(lda >= c.Kb, "conv(bf16): input leading dim %d < packed K %d", lda, c.Kb);
  int Lout = Lin;
  switch (kind) {
    case CONV_K1:
      PT_CHECK(c.slabs == 1, "conv: kernel size %d where 1 expected", c.slabs);
      a.ntaps = 1;
      break;
    case CONV_K3:
    case CONV_K7: {
      const int kt = kind == CONV_K3 ? 3 : 7;
      PT_CHECK(c.slabs == kt, "conv: kernel size %d where %d expected", c.slabs, kt);
      a.ntaps = kt;
      for (int k = 0; k < kt; ++k) { a.tap_d[k] = k - kt / 2; a.tap_w[k] = k; }
      break;
    }
    case CONV_DOWN:
      PT_CHECK(c.slabs == 4 && Lin >= 2, "conv(down): kernel size %d / input length %d", c.slabs, Lin);
      a.ntaps = 4; a.in_stride = 2;
      for (int k = 0; k < 4; ++k) { a.tap_d[k] = k - 1; a.tap_w[k] = k; }
      Lout = (Lin - 2) / 2 + 1;
      break;
    case CONV_UP:
      PT_CHECK(c.slabs == 4, "conv: kernel size %d where 4 expected", c.slabs);
      Lout = 2 * Lin;
      break;
  }
  a.Lout = Lout;
  if (Lout_p) *Lout_p = Lout;
  if (kind != CONV_UP) {
    a.Lgrid = Lout; a.M = B * Lout;
    a.flops = 2.0 * a.M * c.N * c.Cin * a.ntaps * c.merged;
    return launch_gemm_bf16(a, s);
  }
  a.Lgrid = Lin; a.M = B * Lin; a.out_stride = 2; a.ntaps = 2;
  a.flops = 2.0 * a.M * c.N * c.Cin * 2 * c.merged;
  a.out_off = 0; a.tap_d[0] = 0; a.tap_w[0] = 1; a.tap_d[1] = -1; a.tap_w[1] = 3;
  if (launch_gemm_bf16(a, s)) return 1;
  a.out_off = 1; a.tap_d[0] = 0; a.tap_w[0] = 2; a.tap_d[1] = 1; a.tap_w[1] = 0;
  return launch_gemm_bf16(a, s);
}

static int run_stack_b(const ResStack& st, bf16_t*& h, bf16_t*& t1, bf16_t*& t2, int ld, int B, int L, hipStream_t s) {
  if (run_conv_b(st.l0, CONV_K3, h, ld, B, L, nullptr, 0, t1, ld, ACT_LEAKY, 0.2f, nullptr, 0, s)) return 1;
  if (run_conv_b(st.l1, CONV_K3, t1, ld, B, L, nullptr, 0, t2, ld, ACT_LEAKY, 0.2f, nullptr, 0, s)) return 1;
  if (run_conv_b(st.fin, CONV_K3, t2, ld, B, L, nullptr, 0, t1, ld, ACT_LEAKY, 0.0f, h, ld, s)) return 1;
  bf16_t* o = h; h = t1; t1 = o;
  return 0;
}

// Res_CNR_Stack: h -> relu(BN(conv(cnr(cnr(h)))) + h); result lands in t1, buffers rotate.
static int run_stack(const ResStack& st, float*& h, float*& t1, float*& t2, int ld, int B, int L, hipStream_t s) {
  if (run_conv(st.l0, CONV_K3, h, ld, B, L, t1, ld, ACT_LEAKY, 0.2f, nullptr, 0, s)) return 1;
  if (run_conv(st.l1, CONV_K3, t1, ld, B, L, t2, ld, ACT_LEAKY, 0.2f, nullptr, 0, s)) return 1;
  if (run_conv(st.fin, CONV_K3, t2, ld, B, L, t1, ld, ACT_LEAKY, 0.0f, h, ld, s)) return 1;
  float* o = h; h = t1; t1 = o;
  return 0;
}

// ------------------------------------------------------------------ VQ-VAE
static size_t vq_stage_floats(const VQVAE& v, int B, int T, bool decode) {
  // largest rows*ld over the conv stack
  size_t mx = 0;
  if (!decode) {
    int L = T;
    for (int i = 0; i < v.levels; ++i) {
      mx = std::max(mx, (size_t)B * L * ld_of(v.ch[i]));
      if (i < v.levels - 1) L = (L - 2) / 2 + 1;
    }
  } else {
    int L = T;
    for (int i = v.levels - 1; i >= 0; --i) {
      mx = std::max(mx, (size_t)B * L * ld_of(v.ch[i]));
      if (i > 0) { L *= 2; mx = std::max(mx, (size_t)B * L * ld_of(v.ch[i - 1])); }
    }
  }
  return mx;
}

static int latent_len(const VQVAE& v, int T) {
  int L = T;
  for (int i = 0; i < v.levels - 1; ++i) L = (L - 2) / 2 + 1;
  return L;
}

size_t vqvae_workspace_bytes(const Model& m, int B, int T) {
  const VQVAE& v = m.vq;
  // sized for encode(T) and for decode(T' = ceil(T/2^(levels-1))) whichever is larger
  const int Tq = (T + (1 << (v.levels - 1)) - 1) >> (v.levels - 1);
  const size_t enc = (size_t)B * T * ld_of(v.in_dim) + 3 * vq_stage_floats(v, B, T, false) +
                     2 * (size_t)B * std::max(Tq, 1) * ld_of(v.emb_dim);
  const size_t dec = (size_t)B * Tq * ld_of(v.emb_dim) + 3 * vq_stage_floats(v, B, Tq, true) +
                     (size_t)B * (Tq << (v.levels - 1)) * ld_of(v.in_dim);
  return (std::max(enc, dec) * sizeof(float) + 16 * 256 + 255) & ~(size_t)255;
}

int vqvae_encode(Model& m, const float* x, int B, int T, float* z, float* quantized, int64_t* indices, void* ws,
                 size_t ws_bytes, hipStream_t s) {
  PT_CHECK(m.finalized && (m.kind == "vqvae_sc" || m.kind == "vqvae_legacy"), "vqvae_encode: wrong or unfinalized model");
  const VQVAE& v = m.vq;
  PT_CHECK(B > 0 && T >= (1 << (v.levels - 1)), "vqvae_encode: need B>0 and T>=%d (got B=%d T=%d)", 1 << (v.levels - 1), B, T);
  Arena ar(ws, ws_bytes);
  const int ldx = ld_of(v.in_dim), ldz = ld_of(v.emb_dim);
  const size_t stage = vq_stage_floats(v, B, T, false);
  float* x_cl = ar.f32((size_t)B * T * ldx);
  float *h = ar.f32(stage), *t1 = ar.f32(stage), *t2 = ar.f32(stage);
  const int Tq = latent_len(v, T);
  float* z_cl = ar.f32((size_t)B * Tq * ldz);
  float* q_cl = ar.f32((size_t)B * Tq * ldz);
  PT_CHECK(!ar.overflow, "vqvae_encode: workspace too small (%zu bytes given, %zu needed)", ws_bytes, ar.off);
  if (launch_ncl_to_cl(x, x_cl, B, v.in_dim, T, ldx, 0, ldx, s)) return 1;
  int L = T;
  if (run_conv(v.enc_project, CONV_K3, x_cl, ldx, B, L, h, ld_of(v.ch[0]), ACT_LEAKY, 0.2f, nullptr, 0, s)) return 1;
  for (int i = 0; i < v.levels; ++i) {
    const int ld = ld_of(v.ch[i]);
    if (run_stack(v.enc[i], h, t1, t2, ld, B, L, s)) return 1;
    if (i < v.levels - 1) {
      int Lo;
      if (run_conv(v.down[i], CONV_DOWN, h, ld, B, L, t1, ld_of(v.ch[i + 1]), ACT_LEAKY, 0.2f, nullptr, 0, s, &Lo)) return 1;
      float* o = h; h = t1; t1 = o;
      L = Lo;
    }
  }
  if (run_conv(v.pre_vq, CONV_K1, h, ld_of(v.ch[v.levels - 1]), B, L, z_cl, ldz, ACT_NONE, 1.f, nullptr, 0, s)) return 1;
  if (launch_pq_search(z_cl, ldz, v.codebooks, B * L, v.groups, v.emb_dim / v.groups, v.num_codes, indices,
                       quantized ? q_cl : nullptr, ldz, s)) return 1;
  if (z && launch_cl_to_ncl(z_cl, ldz, 0, z, B, v.emb_dim, L, s)) return 1;
  if (quantized && launch_cl_to_ncl(q_cl, ldz, 0, quantized, B, v.emb_dim, L, s)) return 1;
  return 0;
}

int launch_cvt_f32_bf16(const float* in, int ldi, unsigned short* out, int ldo, int rows, int cols, int zero_to,
                        hipStream_t s);   // bf16ops.hip

// bf16 mode of DecoderSC: every activation between the convs is a bf16 channels-last row; the final projection
// writes fp32.  The stage buffers are carved from the same arena sizes as the fp32 path (half used).
static int vqvae_decode_cl_b(Model& m, const float* e_cl, int lde, int B, int Tq, Arena& ar, float** out_cl, int* ld_out,
                             hipStream_t s) {
  const VQVAE& v = m.vq;
  const size_t stage = vq_stage_floats(v, B, Tq, true);
  bf16_t *h = (bf16_t*)ar.raw(stage * 2), *t1 = (bf16_t*)ar.raw(stage * 2), *t2 = (bf16_t*)ar.raw(stage * 2);
  bf16_t* eb = (bf16_t*)ar.raw((size_t)B * Tq * lde * 2);
  const int To = Tq << (v.levels - 1);
  const int ldo = ld_of(v.in_dim);
  float* o_cl = ar.f32((size_t)B * To * ldo);
  PT_CHECK(!ar.overflow, "vqvae_decode: workspace too small (%zu bytes given, %zu needed)", ar.cap, ar.off);
  for (int i = 0; i < v.levels; ++i) PT_CHECK(v.ch[i] % 64 == 0, "vqvae_decode(bf16): channel plan must be multiples of 64");
  PT_CHECK(lde % 64 == 0, "vqvae_decode(bf16): latent width %d must be a multiple of 64", lde);
  if (launch_cvt_f32_bf16(e_cl, lde, eb, lde, B * Tq, lde, lde, s)) return 1;
  int L = Tq;
  if (run_conv_b(v.aft_vq, CONV_K1, eb, lde, B, L, nullptr, 0, h, v.ch[v.levels - 1], ACT_NONE, 1.f, nullptr, 0, s)) return 1;
  for (int i = 0; i < v.levels; ++i) {
    const int lv = v.levels - 1 - i;
    if (run_stack_b(v.dec[i], h, t1, t2, v.ch[lv], B, L, s)) return 1;
    if (i < v.levels - 1) {
      int Lo;
      if (run_conv_b(v.up[i], CONV_UP, h, v.ch[lv], B, L, nullptr, 0, t1, v.ch[lv - 1], ACT_LEAKY, 0.2f, nullptr, 0, s, &Lo)) return 1;
      bf16_t* o = h; h = t1; t1 = o;
      L = Lo;
    }
  }
  if (run_conv_b(v.dec_project, CONV_K1, h, v.ch[0], B, L, o_cl, ldo, nullptr, 0, ACT_NONE, 1.f, nullptr, 0, s, nullptr, ldo)) return 1;
  *out_cl = o_cl;
  *ld_out = ldo;
  return 0;
}

int vqvae_decode_cl(Model& m, const float* e_cl, int lde, int B, int Tq, Arena& ar, float** out_cl, int* ld_out,
                    hipStream_t s) {
  if (m.precision == 1) return vqvae_decode_cl_b(m, e_cl, lde, B, Tq, ar, out_cl, ld_out, s);
  const VQVAE& v = m.vq;
  const size_t stage = vq_stage_floats(v, B, Tq, true);
  float *h = ar.f32(stage), *t1 = ar.f32(stage), *t2 = ar.f32(stage);
  const int To = Tq << (v.levels - 1);
  const int ldo = ld_of(v.in_dim);
  float* o_cl = ar.f32((size_t)B * To * ldo);
  PT_CHECK(!ar.overflow, "vqvae_decode: workspace too small (%zu bytes given, %zu needed)", ar.cap, ar.off);
  int L = Tq;
  if (run_conv(v.aft_vq, CONV_K1, e_cl, lde, B, L, h, ld_of(v.ch[v.levels - 1]), ACT_NONE, 1.f, nullptr, 0, s)) return 1;
  for (int i = 0; i < v.levels; ++i) {
    const int lv = v.levels - 1 - i;
    if (run_stack(v.dec[i], h, t1, t2, ld_of(v.ch[lv]), B, L, s)) return 1;
    if (i < v.levels - 1) {
      int Lo;
      if (run_conv(v.up[i], CONV_UP, h, ld_of(v.ch[lv]), B, L, t1, ld_of(v.ch[lv - 1]), ACT_LEAKY, 0.2f, nullptr, 0, s, &Lo)) return 1;
      float* o = h; h = t1; t1 = o;
      L = Lo;
    }
  }
  if (run_conv(v.dec_project, CONV_K1, h, ld_of(v.ch[0]), B, L, o_cl, ldo, ACT_NONE, 1.f, nullptr, 0, s)) return 1;
  *out_cl = o_cl;
  *ld_out = ldo;
  return 0;
}

int vqvae_decode(Model& m, const float* e_ncl, const void* codes, int codes_i64, int B, int Tq, float* out, void* ws,
                 size_t ws_bytes, hipStream_t s) {
  PT_CHECK(m.finalized && (m.kind == "vqvae_sc" || m.kind == "vqvae_legacy"), "vqvae_decode: wrong or unfinalized model");
  const VQVAE& v = m.vq;
  PT_CHECK(B > 0 && Tq > 0, "vqvae_decode: empty input B=%d T'=%d", B, Tq);
  PT_CHECK((e_ncl != nullptr) != (codes != nullptr), "vqvae_decode: pass exactly one of latent / codes");
  Arena ar(ws, ws_bytes);
  const int lde = ld_of(v.emb_dim);
  float* e_cl = ar.f32((size_t)B * Tq * lde);
  PT_CHECK(!ar.overflow, "vqvae_decode: workspace too small");
  if (e_ncl) {
    if (launch_ncl_to_cl(e_ncl, e_cl, B, v.emb_dim, Tq, lde, 0, lde, s)) return 1;
  } else {
    if (launch_pq_gather(codes, codes_i64, v.codebooks, B * Tq, v.groups, v.emb_dim / v.groups, v.num_codes, e_cl, lde,
                         lde, s)) return 1;
  }
  float* o_cl; int ldo;
  if (vqvae_decode_cl(m, e_cl, lde, B, Tq, ar, &o_cl, &ldo, s)) return 1;
  return launch_cl_to_ncl(o_cl, ldo, 0, out, B, v.in_dim, Tq << (v.levels - 1), s);
}

// ------------------------------------------------------------------ audio condition encoder
size_t cond_encoder_workspace_bytes(const Model& m, int B, int T) {
  const CondEncoder& c = m.ce;
  const size_t f = (size_t)B * T * (ld_of(c.in_dim) + 2 * ld_of(c.hid));
  return (f * sizeof(float) + 8 * 256 + 255) & ~(size_t)255;
}

int cond_encoder_forward(Model& m, const float* aud, int B, int T, float* out_ncl, float* out_cl, int ld_out_cl,
                         int col_off, void* ws, size_t ws_bytes, hipStream_t s) {
  PT_CHECK(m.finalized && m.kind == "cond_encoder", "cond_encoder_forward: wrong or unfinalized model");
  const CondEncoder& c = m.ce;
  PT_CHECK(B > 0 && T >= 8, "cond_encoder_forward: need B>0, T>=8 (got B=%d T=%d)", B, T);
  Arena ar(ws, ws_bytes);
  const int ldi = ld_of(c.in_dim), ld = ld_of(c.hid);
  float* a_cl = ar.f32((size_t)B * T * ldi);
  float* h = ar.f32((size_t)B * T * ld);
  float* t = ar.f32((size_t)B * T * ld);
  PT_CHECK(!ar.overflow, "cond_encoder_forward: workspace too small (%zu given, %zu needed)", ws_bytes, ar.off);
  int L = T;
  if (m.precision == 1) {
    // bf16 mode: activations between the convs are bf16 rows living in the (fp32-sized) stage buffers
    PT_CHECK(ldi % 64 == 0 && ld % 64 == 0, "cond_encoder(bf16): widths %d / %d must be multiples of 64", ldi, ld);
    bf16_t *ab = (bf16_t*)a_cl, *hb = (bf16_t*)h, *tb = (bf16_t*)t;
    if (launch_ncl_to_cl_b(aud, ab, B, c.in_dim, T, ldi, 0, ldi, s)) return 1;
    if (run_conv_b(c.feat, CONV_K1, ab, ldi, B, T, nullptr, 0, hb, ld, ACT_LEAKY, 0.1f, nullptr, 0, s)) return 1;
    for (int i = 0; i < 2; ++i) {
      int Lo;
      if (run_conv_b(c.down[i], CONV_DOWN, hb, ld, B, L, nullptr, 0, tb, ld, ACT_LEAKY, 0.2f, nullptr, 0, s, &Lo)) return 1;
      bf16_t* o = hb; hb = tb; tb = o;
      L = Lo;
    }
    float* dst = out_cl ? out_cl + col_off : (float*)tb;
    const int ldd = out_cl ? ld_out_cl : ld;
    int Lo;
    if (run_conv_b(c.down[2], CONV_DOWN, hb, ld, B, L, dst, ldd, nullptr, 0, ACT_LEAKY, 0.2f, nullptr, 0, s, &Lo, c.hid)) return 1;
    if (out_ncl && launch_cl_to_ncl(dst, ldd, 0, out_ncl, B, c.hid, Lo, s)) return 1;
    return 0;
  }
  if (launch_ncl_to_cl(aud, a_cl, B, c.in_dim, T, ldi, 0, ldi, s)) return 1;
  if (run_conv(c.feat, CONV_K1, a_cl, ldi, B, T, h, ld, ACT_LEAKY, 0.1f, nullptr, 0, s)) return 1;
  for (int i = 0; i < 3; ++i) {
    int Lo;
    const bool last = (i == 2) && out_cl != nullptr;
    float* dst = last ? out_cl + col_off : t;
    if (run_conv(c.down[i], CONV_DOWN, h, ld, B, L, dst, last ? ld_out_cl : ld, ACT_LEAKY, 0.2f, nullptr, 0, s, &Lo,
                 last ? c.hid : -1)) return 1;
    if (!last) { float* o = h; h = t; t = o; }
    L = Lo;
  }
  if (out_ncl) {
    const float* src = out_cl ? out_cl : h;
    if (launch_cl_to_ncl(src, out_cl ? ld_out_cl : ld, out_cl ? col_off : 0, out_ncl, B, c.hid, L, s)) return 1;
  }
  return 0;
}

}  // namespace pt
