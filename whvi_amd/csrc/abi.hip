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

thread_local LaunchNote g_note;

// CU count of the calling thread's current device; the attribute query runs once per device, later calls cost one
// hipGetDevice (a thread-local read) -- the launch-bound training steps make ~100 launches per step
int num_cu()
{
    static int cached[64] = {0};       // written once per device with the same value: benign if two threads race
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

}  // namespace whvi

using namespace whvi;

extern "C" __attribute__((visibility("default"))) int whvi_hip_abi_version(void) { return WHVI_HIP_ABI_VERSION; }
extern "C" __attribute__((visibility("default"))) const char *whvi_last_error(void) { return g_err; }

extern "C" __attribute__((visibility("default"))) int whvi_last_kernel(char *buf, int32_t size)
{
    if (buf == nullptr || size <= 0) return -1;
    if (g_note.family == nullptr) { buf[0] = 0; return 0; }
    int off = snprintf(buf, (size_t)size, "whvi::%s<%s", g_note.family, g_note.type);
    for (int i = 0; i < g_note.n && off > 0 && off < size; ++i)
        off += g_note.is_bool[i] ? snprintf(buf + off, (size_t)(size - off), ", %s", g_note.arg[i] ? "true" : "false")
                                 : snprintf(buf + off, (size_t)(size - off), ", %d", g_note.arg[i]);
    if (off > 0 && off < size) off += snprintf(buf + off, (size_t)(size - off), ">");
    return off;
}

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

namespace whvi {

// ---- F3: reparameterisation + KL of J weight matrices in one launch -----------------------------------
// Replaces, per layer and forward pass, the reference's chain of small ops (src/weights.py:43-64,82-83,92;
// src/utils.py:49-71): softplus(g_rho), g_sigma * eps for every MC sample, the stacking of [g_mu; g_sigma*eps]
// for the fused weight kernel, and the dozen element-wise / reduction launches of kl_diag_normal.
//   sigma[j,i]  = softplus(g_rho[j,i])                       (torch's threshold-20 form)
//   u[j,0,i]    = g_mu[j,i];   u[j,1+k,i] = sigma[j,i] * eps[j,k,i]
//   kl_part[j,b] = sum over the block's i of 0.5*(log(lambda) - log(sigma) - 1 + sigma/lambda + mu*(mu/lambda))
// i.e. the terms of kl_diag_normal(g_mu, g_sigma, 0, lambda) in the reference's argument convention
// (sd1 is a standard deviation, sd2 = lambda a variance -- reproduced as is).
// eps is an INPUT (drawn by torch: injectable for parity tests, graph-safe generator).

// grid = (ceil(D/256), J, ceil(S/8)): blockIdx.z owns 8 MC samples so small-D layers with many samples still
// fill the chip; the z == 0 blocks also write sigma, u[:, 0] and the KL partial sums.
__global__ void __launch_bounds__(256)
reparam_kl_kernel(float *u, float *sigma, float *kl_part, const float *g_mu, const float *g_rho, const float *eps,
                  int S, int D, float lambda_)
{
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool lead = blockIdx.z == 0;
    float term = 0.0f;
    if (i < D) {
        const float mu = g_mu[(size_t)j * D + i], rho = g_rho[(size_t)j * D + i];
        const float sg = rho > 20.0f ? rho : log1pf(expf(rho));
        float *uj = u + (size_t)j * (S + 1) * D + i;
        const float *ej = eps + (size_t)j * S * D + i;
        const int k0 = blockIdx.z * REPARAM_SAMPLES_PER_BLOCK;
        const int k1 = (k0 + REPARAM_SAMPLES_PER_BLOCK < S) ? k0 + REPARAM_SAMPLES_PER_BLOCK : S;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) uj[(size_t)(k + 1) * D] = sg * ej[(size_t)k * D];
        if (lead) {
            sigma[(size_t)j * D + i] = sg;
            uj[0] = mu;
            term = 0.5f * (logf(lambda_) - logf(sg) - 1.0f + sg / lambda_ + mu * (mu / lambda_));
        }
    }
    if (!lead) return;   // block-uniform
    // block reduction: wave shuffle tree, then one add per wave through LDS (deterministic order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) term += __shfl_down(term, off, 64);
    __shared__ float wave_sum[4];
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) kl_part[(size_t)j * gridDim.x + blockIdx.x] = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
}

}  // namespace whvi

extern "C" __attribute__((visibility("default"))) int whvi_reparam_kl_blocks(int64_t D) { return (int)((D + 255) / 256); }

extern "C" __attribute__((visibility("default")))
int whvi_reparam_kl_f32(void *u, void *sigma, void *kl_part, const void *g_mu, const void *g_rho, const void *eps,
                        int64_t J, int64_t S, int64_t D, float lambda_, void *stream)
{
    g_err[0] = 0;
    if (J < 0 || S < 0 || D < 1 || J > 65535 || S > 8 * 65535 || D > (1 << 30))
        return fail(WHVI_ERR_ARG, "whvi_reparam_kl: bad sizes%s (J=%lld, D=%lld)", "", J, D);
    if (J == 0) return WHVI_OK;
    if (!u || !sigma || !kl_part || !g_mu || !g_rho || (S > 0 && !eps))
        return fail(WHVI_ERR_ARG, "whvi_reparam_kl: null pointer%s", "");
    if (!(lambda_ > 0.0f)) return fail(WHVI_ERR_ARG, "whvi_reparam_kl: lambda must be positive%s", "");
    const unsigned gz = (unsigned)((S + REPARAM_SAMPLES_PER_BLOCK - 1) / REPARAM_SAMPLES_PER_BLOCK);
    hipLaunchKernelGGL(reparam_kl_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)J, gz ? gz : 1u), dim3(256), 0,
                       (hipStream_t)stream, (float *)u, (float *)sigma, (float *)kl_part, (const float *)g_mu,
                       (const float *)g_rho, (const float *)eps, (int)S, (int)D, lambda_);
    return after_launch("reparam_kl");
}
