// whvi_amd/csrc/fwht_f64.hip -- batched row FWHT, double storage (one translation unit per dtype so
// the library builds in parallel).  ABI: include/whvi_hip.h.
#include "dispatch.hpp"

extern "C" __attribute__((visibility("hidden")))
int whvi_fwht_variant_f64(void *dst, const void *src, int64_t rows, int32_t log2d, int32_t variant, void *stream)
{
    return whvi::fwht_dispatch<double, false>(dst, src, rows, log2d, variant, stream);
}

extern "C" __attribute__((visibility("default"))) int whvi_fwht_f64(void *dst, const void *src, int64_t rows, int32_t log2d, void *stream)
{
    return whvi::fwht_dispatch<double, false>(dst, src, rows, log2d, 0, stream);
}
