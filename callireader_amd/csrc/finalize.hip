// cr_finalize: derive re-laid-out tensors once the checkpoint tensors are in.
#include "ctx.hpp"

int vit_finalize(cr_ctx* c, hipStream_t st);
int calli_finalize(cr_ctx* c, hipStream_t st);
int llm_finalize(cr_ctx* c, hipStream_t st);

extern "C" int cr_finalize(cr_ctx* c, void* stream) {
    if (!c) return cr_fail(CR_ERR_ARG, "cr_finalize: null context");
    if (c->borrowed) return cr_fail(CR_ERR_STATE, "cr_finalize: this context borrows its weights (cr_share_weights); finalize the owner");
    CR_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    if (c->w.count("vision_model.embeddings.patch_embedding.weight")) CR_TRY(vit_finalize(c, st));
    if (c->w.count("normed_emb.weight")) CR_TRY(calli_finalize(c, st));
    // (w1 / w3 are released once interleaved: a context finalized before still has its language model -- a reloaded wqkv / wo / w2 / LM head
    //  needs its decode-layout copy rebuilt just the same)
    if (c->w.count("language_model.model.layers.0.feed_forward.w1.weight") || c->w.count("derived.w13.0")) CR_TRY(llm_finalize(c, st));
    CR_HIP(hipStreamSynchronize(st));
    c->finalized = true;
    cr_bump_gen(c);
    return CR_OK;
}
