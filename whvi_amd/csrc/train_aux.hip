// whvi_amd/csrc/train_aux.hip -- the small kernels around the weight pipeline in a training step (SURVEY.md
// F1 / F3): backward of the reparameterisation + KL launch, and the Gaussian mean-negative-log-likelihood
// reduction with its backward.  Each replaces a dozen ATen launches over the same tensors.  ABI: include/whvi_hip.h.
#include "dispatch.hpp"

namespace whvi {

// ---- backward of whvi_reparam_kl_f32 (closed form) ------------------------------------------------------
//   u[j,0] = mu, u[j,1+k] = sigma * eps[j,k], kl[j] = sum_i 0.5*(log lam - log sigma - 1 + sigma/lam + mu*(mu/lam))
//   d mu    = gu[j,0]                      + gk[j] * (mu / lam)
//   d sigma = sum_k gu[j,1+k] * eps[j,k]   + gk[j] * 0.5 * (1/lam - 1/sigma)
//   d rho   = d sigma * sigmoid(rho)            (softplus' ; 1 above torch's threshold of 20)
__global__ void __launch_bounds__(256)
reparam_kl_bwd_kernel(float *grad_mu, float *grad_rho, const float *gu, const float *gk, const float *g_mu,
                      const float *g_rho, const float *eps, const float *sigma, int S, int D, float lambda_)
{
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    const size_t p = (size_t)j * D + i;
    const float k = gk != nullptr ? gk[j] : 0.0f;
    const float mu = g_mu[p], rho = g_rho[p], sg = sigma[p];
    const float *guj = gu != nullptr ? gu + (size_t)j * (S + 1) * D + i : nullptr;
    const float *ej = eps + (size_t)j * S * D + i;
    float dmu = k * (mu / lambda_), dsg = k * (0.5f * (1.0f / lambda_ - 1.0f / sg));
    if (guj != nullptr) {
        dmu += guj[0];
        float acc = 0.0f;
#pragma unroll 4
        for (int s = 0; s < S; ++s) acc += guj[(size_t)(s + 1) * D] * ej[(size_t)s * D];
        dsg += acc;
    }
    grad_mu[p] = dmu;
    grad_rho[p] = rho > 20.0f ? dsg : dsg * (1.0f / (1.0f + expf(-rho)));
}

// ---- in-kernel eps for the reparameterisation -----------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; the generator family torch.randn uses on GPUs) keyed by a 64-bit seed,
// counter = (element column, matrix << 16 | sample block, launch offset + call, offset high word); Box-Muller
// turns each 4 x 32 bits into 4 standard normals.  The generator state lives in DEVICE memory,
//     state[0] = seed,  state[1] = launch offset,  state[2] = blocks finished (scratch),
// and the kernel advances it itself: every block bumps state[2] when it is done and the block that finishes
// last moves the offset on (all other blocks have read it by then: they read it before they finish).  Nothing
// about the draw is baked into the launch, so a captured hipGraph draws fresh eps on every replay.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ void normals4(const uint32_t (&x)[4], float (&z)[4])
{
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float u1 = ((float)(x[2 * p] >> 8) + 0.5f) * 5.9604644775390625e-8f;        // (0, 1), 24 bits
        const float u2 = ((float)(x[2 * p + 1] >> 8) + 0.5f) * 5.9604644775390625e-8f;
        const float rad = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        z[2 * p] = rad * cs;
        z[2 * p + 1] = rad * sn;
    }
}

// same grid and outputs as reparam_kl_kernel, plus eps_out (J, S, D): the draw, kept for the backward pass
__global__ void __launch_bounds__(256)
reparam_kl_philox_kernel(float *u, float *sigma, float *kl_part, float *eps_out, const float *g_mu, const float *g_rho,
                         unsigned long long *state, int S, int D, float lambda_)
{
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool lead = blockIdx.z == 0;
    const unsigned long long seed = state[0], offset = state[1];
    float term = 0.0f;
    if (i < D) {
        const float mu = g_mu[(size_t)j * D + i], rho = g_rho[(size_t)j * D + i];
        const float sg = rho > 20.0f ? rho : log1pf(expf(rho));
        float *uj = u + (size_t)j * (S + 1) * D + i;
        float *ej = eps_out + (size_t)j * S * D + i;
        const int k0 = blockIdx.z * REPARAM_SAMPLES_PER_BLOCK;
        float z[REPARAM_SAMPLES_PER_BLOCK];
#pragma unroll
        for (int call = 0; call < REPARAM_SAMPLES_PER_BLOCK / 4; ++call) {
            const unsigned long long off = offset + (unsigned)call;
            uint32_t c[4] = {(uint32_t)i, ((uint32_t)j << 16) | (uint32_t)blockIdx.z, (uint32_t)off, (uint32_t)(off >> 32)};
            philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            float n4[4];
            normals4(c, n4);
#pragma unroll
            for (int q = 0; q < 4; ++q) z[4 * call + q] = n4[q];
        }
#pragma unroll
        for (int q = 0; q < REPARAM_SAMPLES_PER_BLOCK; ++q) {
            const int k = k0 + q;
            if (k < S) {
                ej[(size_t)k * D] = z[q];
                uj[(size_t)(k + 1) * D] = sg * z[q];
            }
        }
        if (lead) {
            sigma[(size_t)j * D + i] = sg;
            uj[0] = mu;
            term = 0.5f * (logf(lambda_) - logf(sg) - 1.0f + sg / lambda_ + mu * (mu / lambda_));
        }
    }
    __shared__ float wave_sum[4];
    if (lead) {   // block-uniform
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) term += __shfl_down(term, off, 64);
        if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = term;
    }
    __syncthreads();                      // every thread of the block has read the generator state by now
    if (threadIdx.x == 0) {
        if (lead) kl_part[(size_t)j * gridDim.x + blockIdx.x] = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
        __threadfence();
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (atomicAdd(reinterpret_cast<unsigned *>(state + 2), 1u) == total - 1) {      // the last block to finish
            state[1] = offset + REPARAM_SAMPLES_PER_BLOCK / 4;
            state[2] = 0;
        }
    }
}

// ---- Gaussian MNLL (src/likelihoods.py:18-29) ------------------------------------------------------------
//   ld(e) = -0.5 z^2 - log(sigma) - 0.5 log(2 pi),  z = (y - y_hat) / sigma
//   mnll  = scale * sum_e ld(e),  scale = -n / (m * n_mc)
// y_hat is indexed through three (size, stride) pairs sorted by decreasing stride, y through the matching
// broadcast strides (0 along the MC axis): the (batch, out, n_mc) view of an (n_mc, batch, out) buffer is read
// in memory order without a copy.  part[b] = {scaled partial sum of ld, partial sum of z^2} per block.
struct Dims3 { int64_t n1, n2, h0, h1, h2, y0, y1, y2; };    // sizes of dims 1, 2; strides of y_hat and y

constexpr float HALF_LOG_2PI = 0.91893853320467274178f;

__device__ __forceinline__ float block_sum_256(float v, float *smem4)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) smem4[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (smem4[0] + smem4[1]) + (smem4[2] + smem4[3]);
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256)
gauss_mnll_kernel(float *part, const float *y, const float *y_hat, const float *sigma, int64_t total, Dims3 d,
                  float scale)
{
    __shared__ float smem4[4];
    const float sg = sigma[0];
    const float inv = 1.0f / sg, cst = logf(sg) + HALF_LOG_2PI;
    float acc_ld = 0.0f, acc_zz = 0.0f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i2 = e % d.n2, q = e / d.n2, i1 = q % d.n1, i0 = q / d.n1;
        const float yh = y_hat[i0 * d.h0 + i1 * d.h1 + i2 * d.h2];
        const float yy = y[i0 * d.y0 + i1 * d.y1 + i2 * d.y2];
        const float z = (yy - yh) * inv;
        acc_zz += z * z;
        acc_ld += -0.5f * (z * z) - cst;
    }
    const float s_ld = block_sum_256(acc_ld, smem4), s_zz = block_sum_256(acc_zz, smem4);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = scale * s_ld;
        part[2 * blockIdx.x + 1] = s_zz;
    }
}

// grad_yhat(e) = g * scale * z / sigma  (written with y_hat's own strides);  block 0 also reduces the forward's
// z^2 partial sums:  grad_sigma = g * scale * (sum z^2 - total) / sigma
__global__ void __launch_bounds__(256)
gauss_mnll_bwd_kernel(float *grad_yhat, float *grad_sigma, const float *g, const float *part, int n_part,
                      const float *y, const float *y_hat, const float *sigma, int64_t total, Dims3 d, float scale)
{
    __shared__ float smem4[4];
    const float sg = sigma[0];
    const float gs = g[0] * scale;
    const float w = gs / (sg * sg);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i2 = e % d.n2, q = e / d.n2, i1 = q % d.n1, i0 = q / d.n1;
        const int64_t o = i0 * d.h0 + i1 * d.h1 + i2 * d.h2;
        grad_yhat[o] = w * (y[i0 * d.y0 + i1 * d.y1 + i2 * d.y2] - y_hat[o]);
    }
    if (blockIdx.x == 0) {
        float zz = 0.0f;
        for (int b = threadIdx.x; b < n_part; b += 256) zz += part[2 * b + 1];
        zz = block_sum_256(zz, smem4);
        if (threadIdx.x == 0) grad_sigma[0] = gs * (zz - (float)total) / sg;
    }
}

inline unsigned mnll_blocks(int64_t total)
{
    const int64_t want = (total + 1023) / 1024;            // ~4 elements per thread
    return (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
}

}  // namespace whvi

using namespace whvi;

extern "C" __attribute__((visibility("default")))
int whvi_reparam_kl_bwd_f32(void *grad_mu, void *grad_rho, const void *grad_u, const void *grad_kl, const void *g_mu,
                            const void *g_rho, const void *eps, const void *sigma, int64_t J, int64_t S, int64_t D,
                            float lambda_, void *stream)
{
    g_err[0] = 0;
    if (J < 0 || S < 0 || D < 1 || J > 65535 || D > (1 << 30))
        return fail(WHVI_ERR_ARG, "whvi_reparam_kl_bwd: bad sizes%s (J=%lld, D=%lld)", "", J, D);
    if (J == 0) return WHVI_OK;
    if (!grad_mu || !grad_rho || !g_mu || !g_rho || !sigma || (S > 0 && grad_u && !eps))
        return fail(WHVI_ERR_ARG, "whvi_reparam_kl_bwd: null pointer%s", "");
    if (!(lambda_ > 0.0f)) return fail(WHVI_ERR_ARG, "whvi_reparam_kl_bwd: lambda must be positive%s", "");
    hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)J), dim3(256), 0,
                       (hipStream_t)stream, (float *)grad_mu, (float *)grad_rho, (const float *)grad_u,
                       (const float *)grad_kl, (const float *)g_mu, (const float *)g_rho, (const float *)eps,
                       (const float *)sigma, (int)S, (int)D, lambda_);
    return after_launch("reparam_kl_bwd");
}

extern "C" __attribute__((visibility("default")))
int whvi_reparam_kl_philox_f32(void *u, void *sigma, void *kl_part, void *eps_out, const void *g_mu, const void *g_rho,
                               void *state, int64_t J, int64_t S, int64_t D, float lambda_, void *stream)
{
    g_err[0] = 0;
    if (J < 0 || S < 0 || D < 1 || J > 65535 || S > 8 * 65535 || D > (1 << 30))
        return fail(WHVI_ERR_ARG, "whvi_reparam_kl_philox: bad sizes%s (J=%lld, D=%lld)", "", J, D);
    if (J == 0) return WHVI_OK;
    if (!u || !sigma || !kl_part || !g_mu || !g_rho || !state || (S > 0 && !eps_out))
        return fail(WHVI_ERR_ARG, "whvi_reparam_kl_philox: null pointer%s", "");
    if ((uintptr_t)state & 7) return fail(WHVI_ERR_ALIGN, "whvi_reparam_kl_philox: %s pointer is not 8-byte aligned", "state");
    if (!(lambda_ > 0.0f)) return fail(WHVI_ERR_ARG, "whvi_reparam_kl_philox: lambda must be positive%s", "");
    const unsigned gz = (unsigned)((S + REPARAM_SAMPLES_PER_BLOCK - 1) / REPARAM_SAMPLES_PER_BLOCK);
    hipLaunchKernelGGL(reparam_kl_philox_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)J, gz ? gz : 1u), dim3(256),
                       0, (hipStream_t)stream, (float *)u, (float *)sigma, (float *)kl_part, (float *)eps_out,
                       (const float *)g_mu, (const float *)g_rho, (unsigned long long *)state, (int)S, (int)D, lambda_);
    return after_launch("reparam_kl_philox");
}

extern "C" __attribute__((visibility("default"))) int whvi_gauss_mnll_blocks(int64_t total) { return (int)mnll_blocks(total); }

static int mnll_check(const int64_t *size, const int64_t *hs, const int64_t *ys, int64_t *total, Dims3 *d)
{
    g_err[0] = 0;
    if (!size || !hs || !ys) return fail(WHVI_ERR_ARG, "whvi_gauss_mnll: null size/stride array%s", "");
    if (size[0] < 0 || size[1] < 0 || size[2] < 0) return fail(WHVI_ERR_ARG, "whvi_gauss_mnll: negative size%s", "");
    *total = size[0] * size[1] * size[2];
    *d = Dims3{size[1] > 0 ? size[1] : 1, size[2] > 0 ? size[2] : 1, hs[0], hs[1], hs[2], ys[0], ys[1], ys[2]};
    return WHVI_OK;
}

extern "C" __attribute__((visibility("default")))
int whvi_gauss_mnll_f32(void *part, const void *y, const void *y_hat, const void *sigma, const int64_t *size,
                        const int64_t *yhat_stride, const int64_t *y_stride, float scale, void *stream)
{
    int64_t total;
    Dims3 d;
    int rc = mnll_check(size, yhat_stride, y_stride, &total, &d);
    if (rc != WHVI_OK) return rc;
    if (!part || !sigma || (total > 0 && (!y || !y_hat))) return fail(WHVI_ERR_ARG, "whvi_gauss_mnll: null pointer%s", "");
    hipLaunchKernelGGL(gauss_mnll_kernel, dim3(mnll_blocks(total)), dim3(256), 0, (hipStream_t)stream, (float *)part,
                       (const float *)y, (const float *)y_hat, (const float *)sigma, total, d, scale);
    return after_launch("gauss_mnll");
}

extern "C" __attribute__((visibility("default")))
int whvi_gauss_mnll_bwd_f32(void *grad_yhat, void *grad_sigma, const void *grad_out, const void *part, const void *y,
                            const void *y_hat, const void *sigma, const int64_t *size, const int64_t *yhat_stride,
                            const int64_t *y_stride, float scale, void *stream)
{
    int64_t total;
    Dims3 d;
    int rc = mnll_check(size, yhat_stride, y_stride, &total, &d);
    if (rc != WHVI_OK) return rc;
    if (!grad_sigma || !grad_out || !part || !sigma || (total > 0 && (!grad_yhat || !y || !y_hat)))
        return fail(WHVI_ERR_ARG, "whvi_gauss_mnll_bwd: null pointer%s", "");
    hipLaunchKernelGGL(gauss_mnll_bwd_kernel, dim3(mnll_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       (float *)grad_yhat, (float *)grad_sigma, (const float *)grad_out, (const float *)part,
                       (int)mnll_blocks(total), (const float *)y, (const float *)y_hat, (const float *)sigma, total, d,
                       scale);
    return after_launch("gauss_mnll_bwd");
}

// ---------------------------------------------------------------------------------------------------------------------
// Learning-rate schedule of the reference's experiments, on the device (src/evaluation.py:25-26: LambdaLR with
// lambda t: lambda0 * (1 + gamma t)^-p, stepped after EVERY batch, src/networks.py:80-81).  One thread: advance the step
// counter and write the rate Adam reads from device memory -- one graph node instead of the seven tiny float64 torch ops
// the same update takes as tensor arithmetic.  float64 like Python's arithmetic in LambdaLR, one rounding to the float32 rate.
namespace whvi {
__global__ void decay_lr_kernel(double *t, float *lr, double base_lr, double lambda0, double gamma, double p, int advance)
{
    double step = *t;
    if (advance) {
        step += 1.0;
        *t = step;
    }
    *lr = (float)(base_lr * (lambda0 * pow(1.0 + gamma * step, -p)));
}
}  // namespace whvi

extern "C" __attribute__((visibility("default")))
int whvi_decay_lr_step(void *t, void *lr, double base_lr, double lambda0, double gamma, double p, int advance, void *stream)
{
    g_err[0] = 0;
    if (!t || !lr) return fail(WHVI_ERR_ARG, "whvi_decay_lr_step: null pointer%s", "");
    if ((uintptr_t)t & 7) return fail(WHVI_ERR_ALIGN, "whvi_decay_lr_step: %s pointer is not 8-byte aligned", "t");
    if ((uintptr_t)lr & 3) return fail(WHVI_ERR_ALIGN, "whvi_decay_lr_step: %s pointer is not 4-byte aligned", "lr");
    hipLaunchKernelGGL(decay_lr_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (double *)t, (float *)lr, base_lr, lambda0,
                       gamma, p, advance);
    return after_launch("decay_lr_step");
}
