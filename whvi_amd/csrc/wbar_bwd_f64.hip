// whvi_amd/csrc/wbar_bwd_f64.hip -- backward of the weight construction, double.  ABI: include/whvi_hip.h.
#include "dispatch.hpp"
#include "wbar_bwd.hpp"

extern "C" __attribute__((visibility("default")))
int whvi_wbar_bwd_f64(void *grad_u, void *part_s1, void *part_s2, const void *grad_w, const void *s1,
                      const void *u, const void *s2, int64_t J, int64_t S, int64_t R, int32_t log2d, int32_t flags,
                      void *stream)
{
    return whvi::wbar_bwd_dispatch<double>(grad_u, part_s1, part_s2, grad_w, s1, u, s2, J, S, R, log2d, flags, stream);
}
