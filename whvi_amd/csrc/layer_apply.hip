// whvi_amd/csrc/layer_apply.hip -- the narrow-input dense product of a stacked layer and the row dot of a transposed column
// layer, all Monte-Carlo samples per launch (f32).  ABI: include/whvi_hip.h (whvi_small_k_apply_f32, whvi_row_dot_f32).
#include "dispatch.hpp"
#include "layer_apply.hpp"

extern "C" __attribute__((visibility("default")))
int whvi_small_k_apply_f32(void *out, const void *x, const void *w, const void *bias, int64_t S, int64_t B, int64_t N,
                           int32_t log2k, int32_t flags, void *stream)
{
    return whvi::small_k_apply_dispatch(out, x, w, bias, S, B, N, log2k, flags, stream);
}

extern "C" __attribute__((visibility("default")))
int whvi_row_dot_f32(void *y, const void *x, const void *w, const void *bias, int64_t S, int64_t B, int32_t log2d, int32_t flags,
                     void *stream)
{
    return whvi::row_dot_dispatch(y, x, w, bias, S, B, log2d, flags, stream);
}
