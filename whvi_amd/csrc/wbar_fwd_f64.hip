// whvi_amd/csrc/wbar_fwd_f64.hip -- weight construction (+ mean matrix add), double.  ABI: include/whvi_hip.h.
#include "dispatch.hpp"
#include "wbar_fwd.hpp"

extern "C" __attribute__((visibility("default")))
int whvi_wbar_fwd_f64(void *dst, const void *s1, const void *u, const void *s2, const void *base, int64_t J, int64_t S,
                      int64_t R, int32_t log2d, int64_t u_group, int64_t u_first, void *stream)
{
    return whvi::wbar_fwd_dispatch<double>(dst, s1, u, s2, base, J, S, R, log2d, u_group, u_first, stream);
}

extern "C" __attribute__((visibility("default")))
int whvi_wbar_fwd_mean_f64(void *dst, const void *s1, const void *u, const void *s2, int64_t J, int64_t S, int64_t R,
                           int32_t log2d, void *stream)
{
    // u is (J, 1 + S, D): row 0 of each group is the mean vector, rows 1 .. S the samples
    return whvi::wbar_fwd_dispatch<double>(dst, s1, u, s2, nullptr, J, S, R, log2d, S + 1, 1, stream, true);
}
