#include "transformer.h"
namespace pt {
int finalize_transformer_models(Model& m) { set_error("kind %s not built yet", m.kind.c_str()); return 1; }
size_t predictor_workspace_bytes(const Model&, int, int) { return 0; }
int predictor_condition(Model&, const float*, const float*, const float*, int, int, float*, void*, size_t, hipStream_t) { set_error("nyi"); return 1; }
int predictor_logits(Model&, const int32_t*, const int64_t*, const float*, int, int, float*, void*, size_t, hipStream_t) { set_error("nyi"); return 1; }
int predictor_sample(Model&, const int64_t*, const float*, const float*, const float*, int, int, int, int32_t*, float*, void*, size_t, hipStream_t) { set_error("nyi"); return 1; }
size_t refiner_workspace_bytes(const Model&, int, int) { return 0; }
int refiner_forward(Model&, const float*, const float*, const float*, const int64_t*, int, int, int, float*, void*, size_t, hipStream_t) { set_error("nyi"); return 1; }
int launch_interp_blend(const float*, int, int, int, int, const float*, const float*, int, float*, hipStream_t) { set_error("nyi"); return 1; }
}
