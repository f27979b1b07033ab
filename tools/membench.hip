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

// Cache-policy and placement experiments on the production geometry (wave = 16 KiB contiguous,
// 1024-thread blocks, all loads -> wait -> all stores).  LD/ST: bit0 sc0, bit1 sc1, bit2 nt.
// MAP: 0 = block b owns tiles [16b, 16b+16); 1 = XCD-contiguous (blocks that share an XCD, b % 8
// equal, walk one contiguous eighth of the buffer); OOP: write to a second buffer.
#define GLD(MODS) asm volatile("global_load_dwordx4 %0, %1, off" MODS : "=v"(c[k]) : "v"(p + k * 64) : "memory")
#define GST(MODS) asm volatile("global_store_dwordx4 %0, %1, off" MODS :: "v"(q + k * 64), "v"(c[k]) : "memory")
template <int LD, int ST, int MAP, bool OOP, int B, int BAR = 0>
__global__ void __launch_bounds__(B) policy_tile(u32x4 *buf, u32x4 *out, int64_t n_tiles)
{
    constexpr int K = 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t b = blockIdx.x;
    if constexpr (MAP == 1) {
        const int64_t nb = gridDim.x, per = nb / 8;
        b = (b % 8) * per + b / 8;            // nb is a multiple of 8 in this benchmark
    }
    const int64_t t = b * (B / 64) + wave;
    if (t >= n_tiles) return;
    const u32x4 *p = buf + t * 64 * K + lane;
    u32x4 *q = (OOP ? out : buf) + t * 64 * K + lane;
    u32x4 c[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if constexpr (LD == 0) GLD("");
        else if constexpr (LD == 4) GLD(" nt");
        else if constexpr (LD == 5) GLD(" sc0 nt");
        else if constexpr (LD == 6) GLD(" sc1 nt");
        else if constexpr (LD == 7) GLD(" sc0 sc1 nt");
        else if constexpr (LD == 2) GLD(" sc1");
        else if constexpr (LD == 1) GLD(" sc0");
        else GLD(" sc0 sc1");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (BAR == 1) __syncthreads();      // all 16 waves of the block store their 256 KiB together
    if constexpr (BAR == 2) {                     // emulate compute skew between waves, then re-align
        for (int i = 0; i < (wave & 3); ++i) __builtin_amdgcn_s_sleep(100);
    }
    if constexpr (BAR == 3) {
        for (int i = 0; i < (wave & 3); ++i) __builtin_amdgcn_s_sleep(100);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        c[k][0] ^= 1u;
        if constexpr (ST == 0) GST("");
        else if constexpr (ST == 4) GST(" nt");
        else if constexpr (ST == 5) GST(" sc0 nt");
        else if constexpr (ST == 6) GST(" sc1 nt");
        else if constexpr (ST == 7) GST(" sc0 sc1 nt");
        else if constexpr (ST == 2) GST(" sc1");
        else if constexpr (ST == 1) GST(" sc0");
        else GST(" sc0 sc1");
    }
}

// K = 1 (1 KiB per wave) with an artificial delay between the load and the store: does the K = 1 advantage
// come from writing a line back right after reading it (DRAM row still open)?
template <int SLEEPS, int NT>
__global__ void __launch_bounds__(64) lag_tile(u32x4 *buf, int64_t n_tiles)
{
    const int64_t t = blockIdx.x;
    if (t >= n_tiles) return;
    u32x4 *p = buf + t * 64 + threadIdx.x;
    u32x4 v = ld<NT>(p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < SLEEPS; ++i) __builtin_amdgcn_s_sleep(127);   // 127 * 64 cycles ~ 3.4 us each
    v[0] ^= 1u;
    st<NT>(p, v);
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

struct Ctx { u32x4 *buf; int64_t chunks; int grid; int lds; u32x4 *out; };
template <int LD, int ST, int MAP, bool OOP, int B, int BAR = 0> static void l_policy(void *p)
{
    Ctx *c = (Ctx *)p; int64_t tiles = c->chunks / (64 * 16);
    hipLaunchKernelGGL((policy_tile<LD, ST, MAP, OOP, B, BAR>), dim3((unsigned)(tiles / (B / 64))), dim3(B), 0, 0,
                       c->buf, c->out, tiles);
}

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
template <int SLEEPS, int NT> static void l_lag(void *p)
{
    Ctx *c = (Ctx *)p; int64_t tiles = c->chunks / 64;
    hipLaunchKernelGGL((lag_tile<SLEEPS, NT>), dim3((unsigned)tiles), dim3(64), 0, 0, c->buf, tiles);
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
    CK(hipMalloc(&c.out, bytes));
    for (int rep = 0; rep < 2; ++rep) {
    RUN("prod-like (nt / sc1nt / XCD / B=1024)", (l_policy<4, 6, 1, false, 1024, 0>), 0);
    RUN("  + barrier before stores", (l_policy<4, 6, 1, false, 1024, 1>), 0);
    RUN("  + wave skew (0-3 x 2.7us) before stores", (l_policy<4, 6, 1, false, 1024, 2>), 0);
    RUN("  + wave skew, then barrier", (l_policy<4, 6, 1, false, 1024, 3>), 0);
    RUN("B=512 prod-like", (l_policy<4, 6, 1, false, 512, 0>), 0);
    RUN("B=512 + barrier", (l_policy<4, 6, 1, false, 512, 1>), 0);
    }
    return 0;
}
