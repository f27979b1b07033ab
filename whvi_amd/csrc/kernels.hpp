#pragma once
// whvi_amd/csrc/kernels.hpp -- gfx950 kernels for the WHVI hot path (C ABI: include/whvi_hip.h).
//
// Replaces src/fwht/cuda/fwht_cuda_kernel.cu + fwht_cuda.cpp of the reference.  Design notes
// are in DESIGN.md; the register-tile butterfly network is in fwht_tile.hpp.
//
// Kernel shape: one WAVEFRONT owns one tile of 64*K 16-byte chunks (16 KiB for f32, K = 16 =
// one D = 4096 row, two D = 2048 rows, ...).  No LDS, no barriers: waves are independent, a
// 256-thread block is just 4 of them.  The grid is persistent (a few blocks per CU) and each
// wave walks tiles with a grid stride, loading tile t+stride into registers before it
// butterflies tile t, so every wave keeps 16 KiB of HBM reads in flight while it computes.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/whvi_hip.h"
#include "fwht_tile.hpp"

namespace whvi {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- storage <-> arithmetic type -----------------------------------------------------------
template <typename T> struct Elem;

template <> struct Elem<float> {
    using acc = float;
    static constexpr int VEC = 4;
    static __device__ __forceinline__ void unpack(const u32x4 &raw, acc (&o)[VEC])
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t u = raw[i];   // by value: bit_cast of a vector-element lvalue reads lane 0
            o[i] = __uint_as_float(u);
        }
    }
    static __device__ __forceinline__ u32x4 pack(const acc (&v)[VEC])
    {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __float_as_uint(v[i]);
        return r;
    }
};

template <> struct Elem<int32_t> {
    using acc = int32_t;
    static constexpr int VEC = 4;
    static __device__ __forceinline__ void unpack(const u32x4 &raw, acc (&o)[VEC])
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (int32_t)raw[i];
    }
    static __device__ __forceinline__ u32x4 pack(const acc (&v)[VEC])
    {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = (uint32_t)v[i];
        return r;
    }
};

template <> struct Elem<double> {
    using acc = double;
    static constexpr int VEC = 2;
    static __device__ __forceinline__ void unpack(const u32x4 &raw, acc (&o)[VEC])
    {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            o[i] = __longlong_as_double((long long)(((uint64_t)raw[2 * i + 1] << 32) | raw[2 * i]));
    }
    static __device__ __forceinline__ u32x4 pack(const acc (&v)[VEC])
    {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint64_t u = __builtin_bit_cast(uint64_t, v[i]);
            r[2 * i] = (uint32_t)u;
            r[2 * i + 1] = (uint32_t)(u >> 32);
        }
        return r;
    }
};

// fp16 / bf16: f32 arithmetic, ONE rounding (RNE) when the row is stored.
template <> struct Elem<__half> {
    using acc = float;
    static constexpr int VEC = 8;
    static __device__ __forceinline__ void unpack(const u32x4 &raw, acc (&o)[VEC])
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __half2float(__ushort_as_half((unsigned short)(raw[i] & 0xFFFFu)));
            o[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(raw[i] >> 16)));
        }
    }
    static __device__ __forceinline__ u32x4 pack(const acc (&v)[VEC])
    {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t lo = __half_as_ushort(__float2half_rn(v[2 * i]));
            uint32_t hi = __half_as_ushort(__float2half_rn(v[2 * i + 1]));
            r[i] = lo | (hi << 16);
        }
        return r;
    }
};

template <> struct Elem<__hip_bfloat16> {
    using acc = float;
    static constexpr int VEC = 8;
    static __device__ __forceinline__ void unpack(const u32x4 &raw, acc (&o)[VEC])
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t u = raw[i];
            o[2 * i] = __uint_as_float(u << 16);
            o[2 * i + 1] = __uint_as_float(u & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ uint32_t rne(float f)
    {
        // plain cast: hipcc emits v_cvt_pk_bf16_f32, which keeps NaN a NaN (MI355X_MICROARCH.md)
        __hip_bfloat16 b = __float2bfloat16(f);
        return (uint32_t)__builtin_bit_cast(unsigned short, b);
    }
    static __device__ __forceinline__ u32x4 pack(const acc (&v)[VEC])
    {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = rne(v[2 * i]) | (rne(v[2 * i + 1]) << 16);
        return r;
    }
};

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const u32x4 *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void st16(u32x4 *p, const u32x4 &v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// ---- batched row FWHT ------------------------------------------------------------------------
// dst/src: n_chunks 16-byte chunks; tile t = chunks [t*64*K, (t+1)*64*K).  Only the last tile
// can be partial; its missing chunks belong to rows that do not exist (rows never straddle
// tiles), so they are read as zero and never stored.
template <typename T, int LOG2D, int K, int POLICY, bool PREFETCH, bool NT>
__global__ void __launch_bounds__(256)
fwht_rows_kernel(u32x4 *dst, const u32x4 *src, int64_t n_chunks, int64_t n_tiles)
{
    using E = Elem<T>;
    using A = typename E::acc;
    constexpr int VEC = E::VEC;
    constexpr int TILE = 64 * K;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wpb = blockDim.x >> 6;
    const int64_t stride = (int64_t)gridDim.x * wpb;
    int64_t t = (int64_t)blockIdx.x * wpb + wave;
    if (t >= n_tiles) return;

    auto load_tile = [&](int64_t tile, u32x4 (&raw)[K]) {
        const int64_t base = tile * TILE;
        const u32x4 *p = src + base + lane;
        if (base + TILE <= n_chunks) {
#pragma unroll
            for (int k = 0; k < K; ++k) raw[k] = ld16<NT>(p + k * 64);
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                u32x4 z = {0u, 0u, 0u, 0u};
                raw[k] = (base + k * 64 + lane < n_chunks) ? ld16<NT>(p + k * 64) : z;
            }
        }
    };

    u32x4 raw[K];
    load_tile(t, raw);
    for (;;) {
        A r[K][VEC];
#pragma unroll
        for (int k = 0; k < K; ++k) E::unpack(raw[k], r[k]);

        const int64_t tn = t + stride;
        if constexpr (PREFETCH) {
            if (tn < n_tiles) load_tile(tn, raw);
        }

        fwht_tile<A, VEC, K, LOG2D, POLICY>(r, lane);

        const int64_t base = t * TILE;
        u32x4 *q = dst + base + lane;
        if (base + TILE <= n_chunks) {
#pragma unroll
            for (int k = 0; k < K; ++k) st16<NT>(q + k * 64, E::pack(r[k]));
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (base + k * 64 + lane < n_chunks) st16<NT>(q + k * 64, E::pack(r[k]));
        }

        if (tn >= n_tiles) break;
        t = tn;
        if constexpr (!PREFETCH) load_tile(t, raw);
    }
}

// Rows shorter than one 16-byte chunk whose total size is not a multiple of 16 bytes leave a
// sub-chunk tail of whole rows; one thread per tail row finishes it.
template <typename T, int LOG2D>
__global__ void fwht_tail_kernel(T *dst, const T *src, int64_t first_row, int64_t rows)
{
    using A = typename Elem<T>::acc;
    constexpr int D = 1 << LOG2D;
    int64_t r = first_row + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    A v[D];
#pragma unroll
    for (int j = 0; j < D; ++j) v[j] = (A)src[r * D + j];
#pragma unroll
    for (int h = 1; h < D; h *= 2)
#pragma unroll
        for (int j = 0; j < D; ++j)
            if ((j & h) == 0) bfly(v[j], v[j | h]);
#pragma unroll
    for (int j = 0; j < D; ++j) dst[r * D + j] = (T)v[j];
}

// ---- fused scale -> FWHT -> scale -> FWHT -> scale ---------------------------------------------
// Same tile ownership as fwht_rows_kernel.  AXIS_COL: scale vectors are indexed by the column
// (idx mod D) and fetched as 16-byte chunks in the same lane layout as the data (L1/L2 hits:
// a and c are D elements shared by every row, b is n_samples*D).  AXIS_ROW: one scalar per row.
// EYE: src is not read; row i of each group is c[i] * e_i (torch.diag(s2), src/weights.py:73).
// Every multiply is its own rounding (built with -ffp-contract=off), like the reference's
// separate matmul_diag_left kernels (src/utils.py:4-12).
template <typename T, int LOG2D, int K, int AXIS, bool EYE>
__global__ void __launch_bounds__(256)
fused_shs_kernel(u32x4 *dst, const u32x4 *src, const T *a, const T *b, const T *c,
                 int64_t n_chunks, int64_t n_tiles, int64_t n_samples, int64_t sample_stride,
                 int64_t group_rows)
{
    using E = Elem<T>;
    using A = typename E::acc;
    constexpr int VEC = E::VEC;
    constexpr int LV = ilog2(VEC);
    constexpr int TILE = 64 * K;
    constexpr int64_t D = (int64_t)1 << LOG2D;
    static_assert(LOG2D >= LV, "fused kernel handles rows of at least one chunk");
    constexpr int CPR = 1 << (LOG2D - LV);   // chunks per row

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wpb = blockDim.x >> 6;
    const int64_t stride = (int64_t)gridDim.x * wpb;

    for (int64_t t = (int64_t)blockIdx.x * wpb + wave; t < n_tiles; t += stride) {
        const int64_t base = t * TILE;
        const bool full = base + TILE <= n_chunks;
        auto chunk_ok = [&](int k) { return full || (base + k * 64 + lane < n_chunks); };
        auto chunk_row = [&](int k) { return (base + k * 64 + lane) >> (LOG2D - LV); };
        auto chunk_col = [&](int k) { return (int)((base + k * 64 + lane) & (CPR - 1)); };
        // scale factors of chunk k: VEC column values (AXIS_COL) or one row scalar broadcast
        auto scale = [&](const T *vec, int64_t vec_base, int k, A (&out)[VEC]) {
            if constexpr (AXIS == WHVI_AXIS_COL) {
                E::unpack(*reinterpret_cast<const u32x4 *>(vec + vec_base + (int64_t)chunk_col(k) * VEC), out);
            } else {
                const A v = (A)vec[vec_base + chunk_row(k) % group_rows];
#pragma unroll
                for (int e = 0; e < VEC; ++e) out[e] = v;
            }
        };

        A r[K][VEC];
        // ---- load (or synthesise) + first scale
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool ok = chunk_ok(k);
            if constexpr (EYE) {
                const int64_t i = chunk_row(k) % group_rows;   // group_rows == D
                const A cv = (c != nullptr) ? (A)c[i] : (A)1;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    r[k][e] = ((int64_t)chunk_col(k) * VEC + e == i) ? cv : (A)0;
            } else {
                u32x4 raw = {0u, 0u, 0u, 0u};
                if (ok) raw = src[base + k * 64 + lane];
                E::unpack(raw, r[k]);
                if (c != nullptr) {
                    A cv[VEC];
                    scale(c, 0, k, cv);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) r[k][e] = cv[e] * r[k][e];
                }
            }
        }
        fwht_tile<A, VEC, K, LOG2D, POLICY_DPP>(r, lane);
        // ---- middle scale (per MC sample)
        if (b != nullptr) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int64_t s = (chunk_row(k) / sample_stride) % n_samples;
                A bv[VEC];
                scale(b, s * (AXIS == WHVI_AXIS_COL ? D : group_rows), k, bv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) r[k][e] = bv[e] * r[k][e];
            }
        }
        fwht_tile<A, VEC, K, LOG2D, POLICY_DPP>(r, lane);
        // ---- last scale + store
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (a != nullptr) {
                A av[VEC];
                scale(a, 0, k, av);
#pragma unroll
                for (int e = 0; e < VEC; ++e) r[k][e] = av[e] * r[k][e];
            }
            if (chunk_ok(k)) dst[base + k * 64 + lane] = E::pack(r[k]);
        }
    }
}

}  // namespace whvi

