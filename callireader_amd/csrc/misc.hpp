#pragma once
#include "common.hpp"

// vit_misc.hip
int launch_im2col14(const bf16* px, bf16* out, int T, hipStream_t stream);
int launch_cls_rows(const bf16* cls, const bf16* pos, bf16* x, int T, int C, int tokens, hipStream_t stream);
