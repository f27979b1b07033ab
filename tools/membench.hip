// tools/membench.hip -- access-structure microbenchmark behind the FWHT kernel's tile design.
// In-place "copy" (load 16 B/lane, store to the same address) of a large buffer with the tile
// geometry of whvi::fwht_rows_kernel as free parameters.  Not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench tools/membench.hip && tools/membench [GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p)
{
    if constexpr (NT & 1) return __builtin_nontemporal_load(p); else return *p;
}
template <int NT> __device__ __forceinline__ void st(u32x4 *p, u32x4 v)
{
    if constexpr (NT & 2) __builtin_nontemporal_store(v, p); else *p = v;
}

// WAVE tiles: wave owns 64*K chunks, lane l chunk k*64+l (the FWHT kernel's layout)
template <int K, bool PREFETCH, int NT, int B = 256, bool ROT = false>
__global__ void __launch_bounds__(B) wave_tile(u32x4 *buf, int64_t n_tiles)
{
    extern __shared__ char lds_pad[];
    if (n_tiles < 0) lds_pad[threadIdx.x] = 0;   // keeps the dynamic LDS allocation alive
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wpb = blockDim.x >> 6, stride = (int64_t)gridDim.x * wpb;
    int64_t t = (int64_t)blockIdx.x * wpb + wave;
    if (t >= n_tiles) return;
    u32x4 r[K];
    if constexpr (ROT) {
        // every wave starts its 16 loads at a different chunk: de-correlates the DRAM channel phase
        const int rot = (int)(t & (K - 1));
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] = ld<NT>(buf + t * 64 * K + ((k + rot) & (K - 1)) * 64 + lane);
#pragma unroll
        for (int k = 0; k < K; ++k) { r[k][0] ^= 1u; st<NT>(buf + t * 64 * K + ((k + rot) & (K - 1)) * 64 + lane, r[k]); }
        return;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = ld<NT>(buf + t * 64 * K + k * 64 + lane);
    for (;;) {
        u32x4 c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { c[k] = r[k]; c[k][0] ^= 1u; }
        const int64_t tn = t + stride;
        if (PREFETCH && tn < n_tiles) {
#pragma unroll
            for (int k = 0; k < K; ++k) r[k] = ld<NT>(buf + tn * 64 * K + k * 64 + lane);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) st<NT>(buf + t * 64 * K + k * 64 + lane, c[k]);
        if (tn >= n_tiles) break;
        t = tn;
        if (!PREFETCH) {
#pragma unroll
            for (int k = 0; k < K; ++k) r[k] = ld<NT>(buf + t * 64 * K + k * 64 + lane);
        }
    }
}

// ROW tiles: a block of B threads owns ROWS_PER_BLOCK "rows" of 16 KiB; every row is spread over
// B/ROWS_PER_BLOCK threads with K = 1024 / that many chunks per thread.  All loads of the block
// complete (wait + optional barrier) before any store is issued -- the dependency structure of a
// transform whose last stage needs the whole row.  SYNC: 1 = per-wave vmcnt(0), 2 = __syncthreads.
template <int K, int B, int NT, int SYNC>
__global__ void __launch_bounds__(B) row_tile(u32x4 *buf, int64_t n_blocks)
{
    const int64_t t = blockIdx.x;
    if (t >= n_blocks) return;
    // thread i of the block: chunks i + k*B  (wave-instruction = 1 KiB contiguous, block-instruction = B*16 B)
    u32x4 c[K];
    u32x4 *base = buf + t * (int64_t)B * K + threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) c[k] = ld<NT>(base + k * B);
    if constexpr (SYNC >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (SYNC >= 2) __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { c[k][0] ^= 1u; st<NT>(base + k * B, c[k]); }
}

// BLOCK tiles: block of B threads owns B*K chunks, thread i chunk k*B+i (elementwise-kernel layout)
template <int K, int B, int NT>
__global__ void __launch_bounds__(B) block_tile(u32x4 *buf, int64_t n_tiles)
{
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        u32x4 c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) c[k] = ld<NT>(buf + t * B * K + k * B + threadIdx.x);
#pragma unroll
        for (int k = 0; k < K; ++k) { c[k][0] ^= 1u; st<NT>(buf + t * B * K + k * B + threadIdx.x, c[k]); }
    }
}

static float time_ms(void (*launch)(void *), void *ctx, int iters)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(ctx); launch(ctx);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch(ctx);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / iters;
}

struct Ctx { u32x4 *buf; int64_t chunks; int grid; int lds; };

template <int K, bool PF, int NT, int B = 256, bool ROT = false> static void l_wave(void *p)
{
    Ctx *c = (Ctx *)p; int64_t tiles = c->chunks / (64 * K);
    constexpr int WPB = B / 64;
    int64_t want = (tiles + WPB - 1) / WPB; int grid = c->grid > 0 && c->grid < want ? c->grid : (int)want;
    hipLaunchKernelGGL((wave_tile<K, PF, NT, B, ROT>), dim3(grid), dim3(B), c->lds, 0, c->buf, tiles);
}
template <int K, int B, int NT, int SYNC> static void l_row(void *p)
{
    Ctx *c = (Ctx *)p; int64_t blocks = c->chunks / ((int64_t)B * K);
    hipLaunchKernelGGL((row_tile<K, B, NT, SYNC>), dim3((unsigned)blocks), dim3(B), c->lds, 0, c->buf, blocks);
}
template <int K, int B, int NT> static void l_block(void *p)
{
    Ctx *c = (Ctx *)p; int64_t tiles = c->chunks / (B * K);
    int grid = c->grid > 0 && c->grid < tiles ? c->grid : (int)tiles;
    hipLaunchKernelGGL((block_tile<K, B, NT>), dim3(grid), dim3(B), 0, 0, c->buf, tiles);
}

int main(int argc, char **argv)
{
    const int64_t gib = argc > 1 ? atoll(argv[1]) : 16;
    const int64_t bytes = gib << 30;
    Ctx c; c.chunks = bytes / 16;
    CK(hipMalloc(&c.buf, bytes));
    CK(hipMemset(c.buf, 1, bytes));
    const double tb = 2.0 * bytes / 1e12;
#define RUN(name, fn, g) do { c.grid = g; float ms = time_ms(fn, &c, 6); printf("%-44s grid %7d : %7.3f ms  %5.2f TB/s\n", name, g, ms, tb / (ms * 1e-3)); fflush(stdout); } while (0)
    c.lds = 0;
    // 16 KiB per wave (what the register-resident D=4096 transform needs), wave-local wait
    RUN("row K=16 B=64   wait-all", (l_row<16, 64, 3, 1>), 0);
    RUN("row K=16 B=256  wait-all", (l_row<16, 256, 3, 1>), 0);
    RUN("row K=16 B=1024 wait-all", (l_row<16, 1024, 3, 1>), 0);
    RUN("row K=16 B=256  no wait", (l_row<16, 256, 3, 0>), 0);
    // 16 KiB per 4 waves (LDS exchange between 4 waves), barrier
    RUN("row K=4  B=256  barrier (1 row/block)", (l_row<4, 256, 3, 2>), 0);
    RUN("row K=4  B=1024 barrier (4 rows/block)", (l_row<4, 1024, 3, 2>), 0);
    // 16 KiB per 16 waves, barrier
    RUN("row K=1  B=1024 barrier (1 row/block)", (l_row<1, 1024, 3, 2>), 0);
    RUN("row K=2  B=512  barrier (1 row/block)", (l_row<2, 512, 3, 2>), 0);
    RUN("row K=2  B=1024 barrier (2 rows/block)", (l_row<2, 1024, 3, 2>), 0);
    RUN("row K=8  B=128  barrier (1 row/block)", (l_row<8, 128, 3, 2>), 0);
    RUN("row K=8  B=256  barrier (2 rows/block)", (l_row<8, 256, 3, 2>), 0);
    RUN("row K=1  B=256  barrier (4 KiB rows)", (l_row<1, 256, 3, 2>), 0);
    RUN("row K=1  B=64   wait-all (1 KiB rows)", (l_row<1, 64, 3, 1>), 0);
    RUN("row K=2  B=64   wait-all (2 KiB rows)", (l_row<2, 64, 3, 1>), 0);
    RUN("row K=4  B=64   wait-all (4 KiB rows)", (l_row<4, 64, 3, 1>), 0);
    RUN("row K=8  B=64   wait-all (8 KiB rows)", (l_row<8, 64, 3, 1>), 0);
    // plain (no NT) versions of the main candidates
    RUN("row K=16 B=256  wait-all plain", (l_row<16, 256, 0, 1>), 0);
    RUN("row K=1  B=1024 barrier plain", (l_row<1, 1024, 0, 2>), 0);
    RUN("row K=4  B=256  barrier plain", (l_row<4, 256, 0, 2>), 0);
    return 0;
}
