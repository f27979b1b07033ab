// whvi_amd/csrc/fused_f64.hip -- fused scale/FWHT/scale/FWHT/scale pipeline, double.  ABI: include/whvi_hip.h.
#include "dispatch.hpp"

extern "C" __attribute__((visibility("default")))
int whvi_fused_shs_ex_f64(void *dst, const void *src, const void *a, const void *b, const void *c,
                          int64_t rows, int32_t log2d, int64_t n_samples, int64_t sample_stride,
                          int64_t group_rows, int32_t axis, int32_t flags, void *stream)
{
    return whvi::fused_dispatch<double>(dst, src, a, b, c, rows, log2d, n_samples, sample_stride, group_rows,
                                    axis, flags, stream);
}

extern "C" __attribute__((visibility("default")))
int whvi_fused_shs_f64(void *dst, const void *src, const void *a, const void *b, const void *c,
                       int64_t rows, int32_t log2d, int64_t n_samples, int64_t sample_stride,
                       int64_t group_rows, int32_t axis, void *stream)
{
    return whvi::fused_dispatch<double>(dst, src, a, b, c, rows, log2d, n_samples, sample_stride, group_rows,
                                    axis, 0, stream);
}
