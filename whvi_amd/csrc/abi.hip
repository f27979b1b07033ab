// whvi_amd/csrc/abi.hip -- dtype-independent part of the C ABI (include/whvi_hip.h).
#include "dispatch.hpp"

namespace whvi {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *a, long long x, long long y)
{
    snprintf(g_err, sizeof(g_err), fmt, a, x, y);
    return code;
}

int after_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "whvi: %s launch failed: %s", what, hipGetErrorString(e));
        return WHVI_ERR_LAUNCH;
    }
    return WHVI_OK;
}

// CU count of the current device, per calling thread's device (cheap attribute query)
int num_cu()
{
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
        return n;
    return 256;
}

}  // namespace whvi

using namespace whvi;

extern "C" __attribute__((visibility("default"))) int whvi_hip_abi_version(void) { return WHVI_HIP_ABI_VERSION; }
extern "C" __attribute__((visibility("default"))) const char *whvi_last_error(void) { return g_err; }

extern "C" __attribute__((visibility("default"))) int whvi_max_log2d(int32_t dtype)
{
    switch (dtype) {
    case WHVI_F32: return max_log2d<float>();
    case WHVI_F64: return max_log2d<double>();
    case WHVI_F16: return max_log2d<__half>();
    case WHVI_BF16: return max_log2d<__hip_bfloat16>();
    case WHVI_I32: return max_log2d<int32_t>();
    default: return -1;
    }
}

// per-dtype implementations live in fwht_<dtype>.hip
extern "C" __attribute__((visibility("hidden"))) int whvi_fwht_variant_f32(void *, const void *, int64_t, int32_t, int32_t, void *);
extern "C" __attribute__((visibility("hidden"))) int whvi_fwht_variant_f64(void *, const void *, int64_t, int32_t, int32_t, void *);
extern "C" __attribute__((visibility("hidden"))) int whvi_fwht_variant_f16(void *, const void *, int64_t, int32_t, int32_t, void *);
extern "C" __attribute__((visibility("hidden"))) int whvi_fwht_variant_bf16(void *, const void *, int64_t, int32_t, int32_t, void *);
extern "C" __attribute__((visibility("hidden"))) int whvi_fwht_variant_i32(void *, const void *, int64_t, int32_t, int32_t, void *);

extern "C" __attribute__((visibility("default"))) int whvi_fwht_ex(void *dst, const void *src, int64_t rows, int32_t log2d, int32_t dtype,
                            int32_t variant, void *stream)
{
    switch (dtype) {
    case WHVI_F32: return whvi_fwht_variant_f32(dst, src, rows, log2d, variant, stream);
    case WHVI_F64: return whvi_fwht_variant_f64(dst, src, rows, log2d, variant, stream);
    case WHVI_F16: return whvi_fwht_variant_f16(dst, src, rows, log2d, variant, stream);
    case WHVI_BF16: return whvi_fwht_variant_bf16(dst, src, rows, log2d, variant, stream);
    case WHVI_I32: return whvi_fwht_variant_i32(dst, src, rows, log2d, variant, stream);
    default: g_err[0] = 0; return fail(WHVI_ERR_ARG, "whvi: unknown dtype%s %lld", "", dtype);
    }
}
