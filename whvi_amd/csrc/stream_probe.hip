// whvi_amd/csrc/stream_probe.hip -- measurement aid: the HBM ceiling of the transform kernels' OWN access pattern.
// An in-place (or out-of-place) copy with the production geometry of fwht_rows_kernel's streaming launch and nothing
// else: 256-thread blocks, one 16 KiB tile per wave, XCD-contiguous block order, 16 non-temporal global loads per lane,
// the block barrier in front of the stores, 16 write-through non-temporal buffer stores from the tile's base with one issue
// slot between them (kernels.hpp: the spacing the headline kernel's stores have).  bench.py times it in the same run and
// prints it next to the spec peak as `roofline.ceiling_measured`: what this memory system gives a kernel that moves the
// same bytes the same way and computes nothing.  The reference has no counterpart.  ABI: include/whvi_hip.h.
#include "dispatch.hpp"

namespace whvi {

template <typename T, int K, int BLOCK>      // (T only names the launch like the kernels it stands in for: whvi_last_kernel prints <type, ...>)
__global__ void __launch_bounds__(BLOCK)
stream_copy_kernel(u32x4 *dst, const u32x4 *src, int64_t n_chunks, int64_t n_tiles)
{
    constexpr int TILE = 64 * K;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) blk = (blk & 7) * (int64_t)(gridDim.x >> 3) + (blk >> 3);
    const int64_t t = blk * (BLOCK / 64) + wave;
    if (t >= n_tiles) {
        __syncthreads();
        return;
    }
    const int64_t base = t * TILE;
    const bool full = base + TILE <= n_chunks;
    u32x4 raw[K];
    const u32x4 *p = src + base + lane;
    if (full) {
#pragma unroll
        for (int k = 0; k < K; ++k) raw[k] = ld16<true>(p + k * 64);
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            raw[k] = u32x4{0u, 0u, 0u, 0u};
            if (base + k * 64 + lane < n_chunks) raw[k] = ld16<true>(p + k * 64);
        }
    }
    __syncthreads();
    if (full) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            tile_store_stream(dst + base, lane, k, raw[k], TILE * 16);
            asm volatile("s_nop 0");
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (base + k * 64 + lane < n_chunks) st16<true>(dst + base + k * 64 + lane, raw[k]);
    }
}

}  // namespace whvi

extern "C" __attribute__((visibility("default")))
int whvi_stream_copy_probe(void *dst, const void *src, int64_t bytes, void *stream)
{
    using namespace whvi;
    g_err[0] = 0;
    if (bytes < 0 || (bytes & 15)) return fail(WHVI_ERR_ARG, "whvi_stream_copy_probe: bytes%s = %lld is not a multiple of 16", "", bytes);
    if (bytes == 0) return WHVI_OK;
    if (!dst || !src) return fail(WHVI_ERR_ARG, "whvi_stream_copy_probe: null pointer%s", "");
    if (((uintptr_t)dst | (uintptr_t)src) & 15) return fail(WHVI_ERR_ALIGN, "whvi_stream_copy_probe: a pointer%s is not 16-byte aligned", "");
    if (dst != src) {
        const char *d = (const char *)dst, *s = (const char *)src;
        if (d < s + bytes && s < d + bytes) return fail(WHVI_ERR_OVERLAP, "whvi_stream_copy_probe: dst and src overlap without being equal%s", "");
    }
    constexpr int K = 16, BLOCK = 256;
    const int64_t n_chunks = bytes / 16, n_tiles = (n_chunks + 64 * K - 1) / (64 * K);
    const int64_t grid = (n_tiles + BLOCK / 64 - 1) / (BLOCK / 64);
    if (grid >= ((int64_t)1 << 31)) return fail(WHVI_ERR_SIZE, "whvi_stream_copy_probe: too large%s", "");
    note_launch<float>("stream_copy_kernel", K, BLOCK);
    hipLaunchKernelGGL((stream_copy_kernel<float, K, BLOCK>), dim3((unsigned)grid), dim3(BLOCK), 0, (hipStream_t)stream,
                       (u32x4 *)dst, (const u32x4 *)src, n_chunks, n_tiles);
    return after_launch("stream_copy_probe");
}
