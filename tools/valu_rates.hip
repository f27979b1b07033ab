// tools/valu_rates.hip -- issue rate of the vector instructions the butterfly kernels are built from, on the chip itself.
// Each test is a loop of 64 copies of ONE instruction over 16 independent destination registers (no dependent issue
// closer than 16 instructions, so DPP / permlane read-after-write hazards never bind), timed with s_memtime inside
// the wave and with HIP events around the launch, run with 1, 2 and 4 waves per SIMD (256-thread blocks, one, two, four
// per CU).  Output per resident-wave count: counter ticks and wall-clock ns per instruction per SIMD.  Read the ns:
// ~1.0 = the cheap class (two clocks per wave64 issue), ~1.8 = four clocks, ~3.5 = eight.  This is what decides between
// instruction forms in fwht_tile.hpp: packed f32 adds against two scalar adds, the half <-> float conversions of the
// 16-bit streams, fused DPP operands against mov_dpp + add.  Log: profiles/r02/valu_rates.log.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rates tools/valu_rates.hip && tools/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// One test = one asm template with %0 = destination (also a source: keeps the value live), %1, %2 = other sources.
// R16 emits it for 16 different destinations; the loop body holds four R16 groups.
#define R16(T) \
    asm volatile(T : "+v"(d0) : "v"(a), "v"(b)); asm volatile(T : "+v"(d1) : "v"(a), "v"(b)); \
    asm volatile(T : "+v"(d2) : "v"(a), "v"(b)); asm volatile(T : "+v"(d3) : "v"(a), "v"(b)); \
    asm volatile(T : "+v"(d4) : "v"(a), "v"(b)); asm volatile(T : "+v"(d5) : "v"(a), "v"(b)); \
    asm volatile(T : "+v"(d6) : "v"(a), "v"(b)); asm volatile(T : "+v"(d7) : "v"(a), "v"(b)); \
    asm volatile(T : "+v"(d8) : "v"(a), "v"(b)); asm volatile(T : "+v"(d9) : "v"(a), "v"(b)); \
    asm volatile(T : "+v"(d10) : "v"(a), "v"(b)); asm volatile(T : "+v"(d11) : "v"(a), "v"(b)); \
    asm volatile(T : "+v"(d12) : "v"(a), "v"(b)); asm volatile(T : "+v"(d13) : "v"(a), "v"(b)); \
    asm volatile(T : "+v"(d14) : "v"(a), "v"(b)); asm volatile(T : "+v"(d15) : "v"(a), "v"(b));

#define KERNEL(NAME, TYPE, T) \
    __global__ void __launch_bounds__(256) NAME(uint64_t *cycles, TYPE *sink, int iters, TYPE seed) \
    { \
        TYPE d0 = seed, d1 = seed, d2 = seed, d3 = seed, d4 = seed, d5 = seed, d6 = seed, d7 = seed; \
        TYPE d8 = seed, d9 = seed, d10 = seed, d11 = seed, d12 = seed, d13 = seed, d14 = seed, d15 = seed; \
        TYPE a = seed, b = seed; \
        __syncthreads(); \
        const uint64_t t0 = __builtin_readcyclecounter(); \
        for (int i = 0; i < iters; ++i) { R16(T) R16(T) R16(T) R16(T) } \
        asm volatile("s_waitcnt lgkmcnt(0)"); \
        const uint64_t t1 = __builtin_readcyclecounter(); \
        TYPE s = d0; s += d1; s += d2; s += d3; s += d4; s += d5; s += d6; s += d7; \
        s += d8; s += d9; s += d10; s += d11; s += d12; s += d13; s += d14; s += d15; \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = s; \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }

// two-register form (the swaps exchange lanes between BOTH operands): 8 independent pairs
#define P8(T) \
    asm volatile(T : "+v"(d0), "+v"(d8)); asm volatile(T : "+v"(d1), "+v"(d9)); \
    asm volatile(T : "+v"(d2), "+v"(d10)); asm volatile(T : "+v"(d3), "+v"(d11)); \
    asm volatile(T : "+v"(d4), "+v"(d12)); asm volatile(T : "+v"(d5), "+v"(d13)); \
    asm volatile(T : "+v"(d6), "+v"(d14)); asm volatile(T : "+v"(d7), "+v"(d15));
#define KERNEL_PAIR(NAME, T) \
    __global__ void __launch_bounds__(256) NAME(uint64_t *cycles, float *sink, int iters, float seed) \
    { \
        float d0 = seed, d1 = seed, d2 = seed, d3 = seed, d4 = seed, d5 = seed, d6 = seed, d7 = seed; \
        float d8 = seed, d9 = seed, d10 = seed, d11 = seed, d12 = seed, d13 = seed, d14 = seed, d15 = seed; \
        __syncthreads(); \
        const uint64_t t0 = __builtin_readcyclecounter(); \
        for (int i = 0; i < iters; ++i) { P8(T) P8(T) P8(T) P8(T) P8(T) P8(T) P8(T) P8(T) } \
        const uint64_t t1 = __builtin_readcyclecounter(); \
        float s = d0; s += d1; s += d2; s += d3; s += d4; s += d5; s += d6; s += d7; \
        s += d8; s += d9; s += d10; s += d11; s += d12; s += d13; s += d14; s += d15; \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = s; \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }

typedef float f32x2 __attribute__((ext_vector_type(2)));

KERNEL(k_add_f32, float, "v_add_f32 %0, %0, %1")
KERNEL(k_fma_f32, float, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fmac_f32, float, "v_fmac_f32 %0, %1, %2")
KERNEL(k_pk_add_f32, f32x2, "v_pk_add_f32 %0, %0, %1")
KERNEL(k_pk_mul_f32, f32x2, "v_pk_mul_f32 %0, %0, %1")
KERNEL(k_pk_fma_f32, f32x2, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL(k_add_f64, double, "v_add_f64 %0, %0, %1")
KERNEL(k_fma_f64, double, "v_fma_f64 %0, %0, %1, %2")
KERNEL(k_mul_f64, double, "v_mul_f64 %0, %0, %1")
KERNEL(k_mov_b32, float, "v_mov_b32 %0, %1")
KERNEL(k_xor_b32, float, "v_xor_b32 %0, %0, %1")
KERNEL(k_add_u32, float, "v_add_u32 %0, %0, %1")
KERNEL(k_lshl_add_u64, double, "v_lshl_add_u64 %0, %1, 2, %0")
KERNEL(k_cvt_f32_f16, float, "v_cvt_f32_f16 %0, %1")
KERNEL(k_cvt_f32_f16_sdwa, float, "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
KERNEL(k_cvt_f16_f32, float, "v_cvt_f16_f32 %0, %1")
KERNEL(k_cvt_pk_f16_f32, float, "v_cvt_pk_f16_f32 %0, %1, %2")
KERNEL(k_cvt_pkrtz_f16_f32, float, "v_cvt_pkrtz_f16_f32 %0, %1, %2")
KERNEL(k_cvt_pk_bf16_f32, float, "v_cvt_pk_bf16_f32 %0, %1, %2")
KERNEL(k_fma_mix_f32, float, "v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]")
KERNEL(k_fma_mixlo_f16, float, "v_fma_mixlo_f16 %0, %1, 1.0, %2 op_sel_hi:[0,0,0]")
KERNEL(k_lshlrev_b32, float, "v_lshlrev_b32 %0, 16, %1")
KERNEL(k_and_b32, float, "v_and_b32 %0, 0xffff0000, %1")
KERNEL(k_or_b32_sdwa, float, "v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
KERNEL(k_perm_b32, float, "v_perm_b32 %0, %1, %2, %0")
KERNEL(k_mov_dpp_quad, float, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_mov_dpp_ror8, float, "v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf")
KERNEL(k_add_dpp_quad, float, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_fmac_dpp_quad, float, "v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_fmac_dpp_mirror, float, "v_fmac_f32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0xf")
KERNEL(k_fmac_dpp_ror8, float, "v_fmac_f32_dpp %0, %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf")
KERNEL_PAIR(k_permlane16_swap, "v_permlane16_swap_b32 %0, %1")
KERNEL_PAIR(k_permlane32_swap, "v_permlane32_swap_b32 %0, %1")
KERNEL(k_ds_swizzle, float, "ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,16)")
KERNEL(k_ds_bpermute, float, "ds_bpermute_b32 %0, %1, %2")

template <typename T>
static void run(const char *name, void (*k)(uint64_t *, T *, int, T), uint64_t *cyc, void *sink, int cus, double elems)
{
    const int iters = 2000;
    printf("%-22s", name);
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        const int blocks = cus * waves_per_simd;            // 256-thread blocks: one wave per SIMD each
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, cyc, (T *)sink, iters, T{});
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, cyc, (T *)sink, iters, T{});
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * blocks * 4);
        CK(hipMemcpy(h, cyc, sizeof(uint64_t) * blocks * 4, hipMemcpyDeviceToHost));
        double sum = 0;
        for (int i = 0; i < blocks * 4; ++i) sum += (double)h[i];
        free(h);
        const double per_wave = sum / (blocks * 4) / (iters * 64.0);     // clocks per instruction seen by one wave
        // counter ticks (s_memtime: may be a fixed-rate counter, read the ratio to v_add_f32) and wall-clock ns
        printf("  %dw: %6.2f ticks %6.3f ns", waves_per_simd, per_wave / waves_per_simd,
               ms * 1e6 / (iters * 64.0 * waves_per_simd));
    }
    printf("   (%g f32 lanes-worth per inst)\n", elems);
}

int main()
{
    int dev, cus, clk;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev));
    printf("CUs %d, clock %d kHz; columns: counter ticks and ns per instruction per SIMD with 1 / 2 / 4 resident waves per SIMD\n", cus, clk);
    uint64_t *cyc;
    void *sink;
    CK(hipMalloc(&cyc, sizeof(uint64_t) * cus * 4 * 4 * 4));
    CK(hipMalloc(&sink, 8 * 256 * (size_t)cus * 4));
#define RUN(K, T, E) run<T>(#K, K, cyc, sink, cus, E)
    RUN(k_add_f32, float, 1); RUN(k_fma_f32, float, 1); RUN(k_fmac_f32, float, 1);
    RUN(k_pk_add_f32, f32x2, 2); RUN(k_pk_mul_f32, f32x2, 2); RUN(k_pk_fma_f32, f32x2, 2);
    RUN(k_add_f64, double, 1); RUN(k_fma_f64, double, 1); RUN(k_mul_f64, double, 1);
    RUN(k_mov_b32, float, 1); RUN(k_xor_b32, float, 1); RUN(k_add_u32, float, 1); RUN(k_lshl_add_u64, double, 1);
    RUN(k_cvt_f32_f16, float, 1); RUN(k_cvt_f32_f16_sdwa, float, 1); RUN(k_cvt_f16_f32, float, 1);
    RUN(k_cvt_pk_f16_f32, float, 2); RUN(k_cvt_pkrtz_f16_f32, float, 2); RUN(k_cvt_pk_bf16_f32, float, 2);
    RUN(k_fma_mix_f32, float, 1); RUN(k_fma_mixlo_f16, float, 1);
    RUN(k_lshlrev_b32, float, 1); RUN(k_and_b32, float, 1); RUN(k_or_b32_sdwa, float, 1); RUN(k_perm_b32, float, 1);
    RUN(k_mov_dpp_quad, float, 1); RUN(k_mov_dpp_ror8, float, 1); RUN(k_add_dpp_quad, float, 1);
    RUN(k_fmac_dpp_quad, float, 1); RUN(k_fmac_dpp_mirror, float, 1); RUN(k_fmac_dpp_ror8, float, 1);
    RUN(k_permlane16_swap, float, 2); RUN(k_permlane32_swap, float, 2);
    RUN(k_ds_swizzle, float, 1); RUN(k_ds_bpermute, float, 1);
    return 0;
}
