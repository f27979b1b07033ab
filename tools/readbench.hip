// tools/readbench.hip -- read-only stream microbenchmark behind whvi::wbar_bwd_kernel's launch geometry: every wave
// reads 16 KiB tiles (lane l, chunk k * 64 + l: the kernels' layout), does a configurable amount of dependent VALU
// work on them and writes ONE value per tile.  Free parameters: one tile per wave vs persistent grid with register
// prefetch, non-temporal vs cached loads, dynamic LDS per block (caps the waves per CU like the LDS-staged transform
// does), VALU instructions per tile.  Not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 -o tools/readbench tools/readbench.hip && tools/readbench [GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}

// incompressible, non-zero contents (small normal floats): rules out any data-dependent shortcut in the memory system
__global__ void fill_random(uint32_t *p, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = 0x3c000000u | (h & 0x007fffffu) | (h & 0x80000000u);      // +/- [2^-7, 2^-6)
    }
}

template <int K, bool PIPE, bool NT, int WORK>
__global__ void __launch_bounds__(256) read_tiles(float *out, const u32x4 *buf, int64_t n_tiles)
{
    extern __shared__ char lds_pad[];
    if (n_tiles < 0) lds_pad[threadIdx.x] = 0;   // keeps the dynamic LDS allocation alive
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t t = (int64_t)blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    u32x4 r[K];
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = ld<NT>(buf + t * 64 * K + k * 64 + lane);
    for (;;) {
        float v[K * 4];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k * 4 + e] = __uint_as_float(r[k][e]);
        const int64_t tn = t + stride;
        if (PIPE && tn < n_tiles) {
#pragma unroll
            for (int k = 0; k < K; ++k) r[k] = ld<NT>(buf + tn * 64 * K + k * 64 + lane);
        }
        // WORK rounds of K * 4 dependent-per-element VALU ops
#pragma unroll
        for (int w = 0; w < WORK; ++w)
#pragma unroll
            for (int i = 0; i < K * 4; ++i) v[i] = v[i] * 1.0001f + v[(i + 1) & (K * 4 - 1)];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < K * 4; ++i) acc += v[i];
        // cross-lane sum: every lane's loads and work feed the stored value (or the compiler sinks them under lane == 0)
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0) out[t] = acc;
        if (!PIPE || tn >= n_tiles) break;
        t = tn;
    }
}

template <int K, bool PIPE, bool NT, int WORK>
static void run(const char *name, float *out, u32x4 *buf, int64_t bytes, int lds_per_block, int blocks_per_cu)
{
    const int64_t n_tiles = bytes / (16 * 64 * K);
    int dev, cus;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t grid = PIPE ? (int64_t)cus * blocks_per_cu : (n_tiles + 3) / 4;
    if (lds_per_block > 65536)
        CK(hipFuncSetAttribute((const void *)read_tiles<K, PIPE, NT, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_per_block));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((read_tiles<K, PIPE, NT, WORK>), dim3((unsigned)grid), dim3(256), lds_per_block, 0, out, buf, n_tiles);
    CK(hipEventRecord(e0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((read_tiles<K, PIPE, NT, WORK>), dim3((unsigned)grid), dim3(256), lds_per_block, 0, out, buf, n_tiles);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s lds/block %6d  %8.4f ms  %7.1f GB/s\n", name, lds_per_block, ms / iters, bytes / (ms / iters * 1e-3) / 1e9);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 1.0;
    const int64_t bytes = (int64_t)(gib * (1 << 30));
    u32x4 *buf;
    float *out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, bytes / 4096 * 4 + 4096));            // one float per tile of the SMALLEST tile size used below (4 KiB)
    hipLaunchKernelGGL(fill_random, dim3(65536), dim3(256), 0, 0, (uint32_t *)buf, bytes / 4);
    CK(hipDeviceSynchronize());
    printf("read-only stream, %.2f GiB, 16 KiB tiles per wave, 256-thread blocks\n", gib);
    for (int lds : {0, 66560}) {        // 66.5 KB per block = the LDS-staged transform's four slabs: 2 blocks per CU
        run<16, false, true, 0>("one tile per wave, nt, no work", out, buf, bytes, lds, 0);
        run<16, false, false, 0>("one tile per wave, cached, no work", out, buf, bytes, lds, 0);
        run<16, false, true, 4>("one tile per wave, nt, 256 VALU", out, buf, bytes, lds, 0);
        run<16, false, true, 16>("one tile per wave, nt, 1024 VALU", out, buf, bytes, lds, 0);
        run<16, false, true, 32>("one tile per wave, nt, 2048 VALU", out, buf, bytes, lds, 0);
        run<16, true, true, 0>("persistent x2/CU + prefetch, nt, no work", out, buf, bytes, lds, 2);
        run<16, true, true, 16>("persistent x2/CU + prefetch, nt, 1024 VALU", out, buf, bytes, lds, 2);
        run<16, true, false, 16>("persistent x2/CU + prefetch, cached, 1024 VALU", out, buf, bytes, lds, 2);
        if (lds == 0) {
            run<16, true, true, 16>("persistent x4/CU + prefetch, nt, 1024 VALU", out, buf, bytes, lds, 4);
            run<16, true, true, 0>("persistent x4/CU + prefetch, nt, no work", out, buf, bytes, lds, 4);
        }
    }
    run<8, false, true, 16>("8 KiB tiles, one per wave, nt, 512 VALU", out, buf, bytes, 0, 0);
    run<4, false, true, 16>("4 KiB tiles, one per wave, nt, 256 VALU", out, buf, bytes, 0, 0);
    return 0;
}
