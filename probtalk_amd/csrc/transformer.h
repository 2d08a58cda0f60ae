// Predictor / refiner module sequences (transformer.hip) and their row-wise kernels (rowops.hip,
// attention.hip).
#pragma once
#include "model.h"

namespace pt {

size_t predictor_workspace_bytes(const Model& m, int B, int T);
int predictor_condition(Model& m, const float* masked_motion, const float* mask, const float* audio, int B, int T,
                        float* condition, void* ws, size_t ws_bytes, hipStream_t s);
int predictor_logits(Model& m, const int32_t* ids, const int64_t* label, const float* condition, int B, int Tq,
                     float* logits, void* ws, size_t ws_bytes, hipStream_t s);
int predictor_sample(Model& m, const int64_t* label, const float* condition, const float* noise_q, const float* noise_g,
                     int steps, int B, int Tq, int32_t* codes_out, float* logits_out, void* ws, size_t ws_bytes, hipStream_t s,
                     unsigned long long seed = 0, const unsigned long long* seed_dev = nullptr);
size_t refiner_workspace_bytes(const Model& m, int B, int T);
int refiner_forward(Model& m, const float* x, const float* condition, const float* mask, const int64_t* label,
                    int n_label, int B, int T, float* out, void* ws, size_t ws_bytes, hipStream_t s);
int launch_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                     int B, int Nq, int Nk, int H, hipStream_t s);
int launch_interp_blend(const float* pred, int B, int C, int T_in, int T_out, const float* gt, const float* mask,
                        int premask, float* out, hipStream_t s);

}  // namespace pt
