// Host-side runtime: weight store, BatchNorm folding / repacking, workspace arena and the
// per-module kernel sequences.  No allocation and no synchronisation happens in any *_forward path.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "pt_common.h"

namespace pt {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> f;    // fp32 payload (dtype 0)
  std::vector<double> d;   // fp64 payload (dtype 1)
  int dtype = 0;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

// Packed, device-resident weights of one tap-GEMM.
struct DevConv {
  float* w = nullptr;   // [slabs][Npad][K]
  float* b = nullptr;   // [Npad]
  int N = 0, Npad = 0, K = 0, Cin = 0, slabs = 0;
  int merged = 1;       // 2 when a residual conv was folded in (the reference runs two convs)
  unsigned short* wb = nullptr;  // bf16 copy [slabs][Npad][Kb] (bf16-mode models)
  int Kb = 0;                    // round_up(Cin, 64)
  unsigned short* wbt = nullptr; // N = 512 Linears: the same weights K-tile-major, [Kb / 32][512][32] (gemm_bf16_row.hip)
};

enum ConvKind { CONV_K1, CONV_K3, CONV_K7, CONV_DOWN, CONV_UP };

struct ResStack {
  DevConv l0, l1, fin;
};

struct VQVAE {
  int in_dim = 0, emb_dim = 0, num_codes = 0, groups = 1, levels = 4;
  std::vector<int> ch;            // encoder channel plan per level
  DevConv enc_project, pre_vq, aft_vq, dec_project;
  std::vector<ResStack> enc, dec;
  std::vector<DevConv> down, up;
  float* codebooks = nullptr;     // [G][K][D]
};

struct CondEncoder {
  int in_dim = 0, hid = 0;
  DevConv feat;
  DevConv down[3];
};

struct Model {
  std::string kind;
  std::vector<int> cfg;
  std::map<std::string, HostTensor> sd;
  std::vector<void*> allocs;
  bool finalized = false;
  int precision = 0;    // 0 = fp32 everywhere; 1 = bf16 operands in the transformer GEMMs / attention
  VQVAE vq;
  CondEncoder ce;
  struct Predictor* pred = nullptr;
  struct Refiner* ref = nullptr;
  ~Model();
};

// Bump allocator over the caller's workspace (256-byte granules).
struct Arena {
  char* base;
  size_t cap, off = 0;
  bool overflow = false;
  Arena(void* p, size_t n) : base((char*)p), cap(n) {}
  float* f32(size_t n) { return (float*)raw(n * sizeof(float)); }
  void* raw(size_t bytes) {
    size_t o = (off + 255) & ~(size_t)255;
    if (base == nullptr || o + bytes > cap) {
      overflow = true;
      off = o + bytes;
      return nullptr;
    }
    off = o + bytes;
    return base + o;
  }
};

int finalize_model(Model& m);

// GEMM wrappers used by the module sequences.
int run_conv(const DevConv& c, ConvKind kind, const float* A, int lda, int B, int Lin, float* C, int ldc, int act,
             float slope, const float* Res, int ldr, hipStream_t s, int* Lout = nullptr, int zero_to = -1,
             const double* pe = nullptr, int pe_period = 0);

typedef unsigned short bf16_t;
int run_conv_b(const DevConv& c, ConvKind kind, const bf16_t* A, int lda, int B, int Lin, float* C, int ldc, bf16_t* Cb,
               int ldcb, int act, float slope, const bf16_t* ResB, int ldr, hipStream_t s, int* Lout = nullptr,
               int zero_to = -1);

size_t vqvae_workspace_bytes(const Model& m, int B, int T);
int vqvae_encode(Model& m, const float* x, int B, int T, float* z, float* quantized, int64_t* indices, void* ws,
                 size_t ws_bytes, hipStream_t s);
int vqvae_decode(Model& m, const float* e_ncl, const void* codes, int codes_i64, int B, int Tq, float* out, void* ws,
                 size_t ws_bytes, hipStream_t s);
// decoder body on a channels-last latent already in the workspace; returns the channels-last output
int vqvae_decode_cl(Model& m, const float* e_cl, int lde, int B, int Tq, Arena& ar, float** out_cl, int* ld_out,
                    hipStream_t s);

size_t cond_encoder_workspace_bytes(const Model& m, int B, int T);
int cond_encoder_forward(Model& m, const float* aud, int B, int T, float* out_ncl, float* out_cl, int ld_out_cl,
                         int col_off, void* ws, size_t ws_bytes, hipStream_t s);

}  // namespace pt
