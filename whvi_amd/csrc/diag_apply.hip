// whvi_amd/csrc/diag_apply.hip -- the as-written weight matrix applied as its diagonal (+ backward), f32 / f64.
// ABI: include/whvi_hip.h (whvi_diag_apply_*).
#include "dispatch.hpp"
#include "diag_apply.hpp"

#define WHVI_EXPORT extern "C" __attribute__((visibility("default")))

WHVI_EXPORT int whvi_diag_apply_f32(void *out, const void *x, const void *s1, const void *s2, const void *u, const void *bias,
                                    int64_t S, int64_t B, int32_t log2d, int32_t flags, void *stream)
{
    return whvi::diag_apply_dispatch<float>(out, x, s1, s2, u, bias, S, B, log2d, flags, stream);
}

WHVI_EXPORT int whvi_diag_apply_f64(void *out, const void *x, const void *s1, const void *s2, const void *u, const void *bias,
                                    int64_t S, int64_t B, int32_t log2d, int32_t flags, void *stream)
{
    return whvi::diag_apply_dispatch<double>(out, x, s1, s2, u, bias, S, B, log2d, flags, stream);
}

WHVI_EXPORT int64_t whvi_diag_apply_bwd_slabs(int32_t dtype, int64_t S, int64_t B, int32_t log2d)
{
    if (dtype == WHVI_F32) return whvi::diag_apply_bwd_slabs_for<float>(S, B, log2d);
    if (dtype == WHVI_F64) return whvi::diag_apply_bwd_slabs_for<double>(S, B, log2d);
    return -1;
}

WHVI_EXPORT int whvi_diag_apply_bwd_f32(void *grad_x, void *out, void *part, const void *g, const void *x, const void *s1,
                                        const void *s2, const void *u, const void *bias, int64_t S, int64_t B, int32_t log2d,
                                        int64_t n_slabs, int32_t flags, void *stream)
{
    return whvi::diag_apply_bwd_dispatch<float>(grad_x, out, part, g, x, s1, s2, u, bias, S, B, log2d, n_slabs, flags, stream);
}

WHVI_EXPORT int whvi_diag_apply_bwd_f64(void *grad_x, void *out, void *part, const void *g, const void *x, const void *s1,
                                        const void *s2, const void *u, const void *bias, int64_t S, int64_t B, int32_t log2d,
                                        int64_t n_slabs, int32_t flags, void *stream)
{
    return whvi::diag_apply_bwd_dispatch<double>(grad_x, out, part, g, x, s1, s2, u, bias, S, B, log2d, n_slabs, flags, stream);
}
