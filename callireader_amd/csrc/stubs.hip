// Entry points whose kernels are not written yet: fail loudly.
#include "ctx.hpp"
int calli_finalize(cr_ctx*, hipStream_t) { return cr_fail(CR_ERR_STATE, "calli stage not built yet"); }
int llm_finalize(cr_ctx*, hipStream_t) { return cr_fail(CR_ERR_STATE, "llm stage not built yet"); }
extern "C" {
int cr_resample(cr_ctx*, const void*, int, void*, void*) { return cr_fail(CR_ERR_STATE, "cr_resample: not built yet"); }
int cr_vq(cr_ctx*, const void*, int, int64_t*, void*, void*) { return cr_fail(CR_ERR_STATE, "cr_vq: not built yet"); }
int cr_denorm(cr_ctx*, const void*, const int64_t*, const void*, int, int, void*, int32_t*, void*) { return cr_fail(CR_ERR_STATE, "cr_denorm: not built yet"); }
int cr_embed_splice(cr_ctx*, const int64_t*, int, const void*, int, int64_t, const void*, int, int64_t, void*, void*) { return cr_fail(CR_ERR_STATE, "cr_embed_splice: not built yet"); }
int cr_kv_alloc(cr_ctx*, int, int, cr_kv**) { return cr_fail(CR_ERR_STATE, "cr_kv_alloc: not built yet"); }
int cr_kv_free(cr_kv*) { return CR_OK; }
int cr_kv_length(const cr_kv*, int) { return 0; }
int cr_kv_reset(cr_kv*, int) { return CR_OK; }
int cr_llm_prefill(cr_ctx*, cr_kv*, int, const void*, int, float*, int64_t*, void*) { return cr_fail(CR_ERR_STATE, "cr_llm_prefill: not built yet"); }
int cr_llm_decode(cr_ctx*, cr_kv*, const int32_t*, int, const int64_t*, float*, float, const int64_t*, int, const int32_t*, int64_t*, void*) { return cr_fail(CR_ERR_STATE, "cr_llm_decode: not built yet"); }
}
