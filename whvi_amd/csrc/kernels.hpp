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
#include <stdlib.h>
#include <string.h>

#include "../../include/whvi_hip.h"
#include "tuning.hpp"
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
        // The high half goes through an explicit shift.  Written as plain C++ the compiler folds shift + convert into
        // v_cvt_f32_f16_sdwa (src0_sel:WORD_1), and the stream runs 2.6 % slower: 6.21 vs 6.37 TB/s at D = 4096, 2^20 rows,
        // interleaved A/B on identical finite data (tools/probe_f16_convert.py, profiles/r02/f16_convert_ab.log; with
        // bf16's bit moves in place of the converts -- wrong values, timing only -- 6.41).  Same values either way.
#if WHVI_F16_UNPACK == 1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float lo, hi;
            uint32_t sh;
            asm("v_cvt_f32_f16 %0, %1" : "=v"(lo) : "v"(raw[i]));
            asm("v_lshrrev_b32 %0, 16, %1" : "=v"(sh) : "v"(raw[i]));
            asm("v_cvt_f32_f16 %0, %1" : "=v"(hi) : "v"(sh));
            o[2 * i] = lo;
            o[2 * i + 1] = hi;
        }
#elif defined(WHVI_TUNING_BUILD) && WHVI_F16_UNPACK == 2   /* timing experiment only: bf16's bit moves (WRONG values) */
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t u = raw[i];
            o[2 * i] = __uint_as_float(u << 16);
            o[2 * i + 1] = __uint_as_float(u & 0xFFFF0000u);
        }
#elif defined(WHVI_TUNING_BUILD) && WHVI_F16_UNPACK == 3   /* the production form as volatile asm */
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float lo, hi;
            uint32_t sh;
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(lo) : "v"(raw[i]));
            asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(sh) : "v"(raw[i]));
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(hi) : "v"(sh));
            o[2 * i] = lo;
            o[2 * i + 1] = hi;
        }
#elif defined(WHVI_TUNING_BUILD)   /* 0: the compiler's form (SDWA operand for the high half) */
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __half2float(__ushort_as_half((unsigned short)(raw[i] & 0xFFFFu)));
            o[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(raw[i] >> 16)));
        }
#else
#error "WHVI_F16_UNPACK: the shipped library has one form (1)"
#endif
    }
    static __device__ __forceinline__ u32x4 pack(const acc (&v)[VEC])
    {
#if defined(WHVI_TUNING_BUILD) && defined(WHVI_F16_PACK_EXP) && WHVI_F16_PACK_EXP == 1   /* timing experiment only: bf16's pack (WRONG values) */
        u32x4 rr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t w;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
            rr[i] = w;
        }
        return rr;
#endif
        // <2 x float> -> <2 x half> fptrunc (round to nearest even) selects gfx950's v_cvt_pk_f16_f32:
        // one instruction per output dword instead of two converts and an or
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2_t p = {v[2 * i], v[2 * i + 1]};
            const f16x2_t h = __builtin_convertvector(p, f16x2_t);
            r[i] = __builtin_bit_cast(uint32_t, h);
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
#if WHVI_BF16_PACK == 1
        // one v_cvt_pk_bf16_f32 per output dword (low half <- first operand).  From `rne(lo) | rne(hi) << 16` the compiler
        // makes TWO of them (upper lane unused) plus a v_or_b32_sdwa: 192 instead of 64 instructions per tile.
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t w;
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
            r[i] = w;
        }
#else
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = rne(v[2 * i]) | (rne(v[2 * i + 1]) << 16);
#endif
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

// Streaming store of a whole tile: buffer stores with the write-through (sc1) + non-temporal policy.
// Measured on the 16 GiB in-place workload (tools/probe_exp.py, interleaved A/B in one process):
// "sc1 nt" stores 6.32 TB/s vs plain "nt" 6.14 -- the line is not kept in L2 at all -- and 6.39 TB/s
// together with the XCD-contiguous block order below.  The descriptor is built from wave-uniform
// values (tile base), one per tile; aux = cache-policy bits (bit 1 nt, bit 4 sc1 on gfx940+).
__device__ __forceinline__ void tile_store_stream(u32x4 *tile_base, int lane, int k, const u32x4 &v, int tile_bytes)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile_base, 0, tile_bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, lane * 16 + k * 1024, 0, (1 << 4) | (1 << 1));
}

// 16-byte load at (wave-uniform base pointer) + lane * 16 + byte offset: a raw buffer load with the base in SGPRs -- no
// 64-bit VGPR address arithmetic per chunk and no address registers (the global_load form of a per-chunk pointer costs a
// v_add_co / v_addc pair and two VGPRs each).  `bytes` bounds the access (reads beyond it return zeros).
template <bool NT = false>
__device__ __forceinline__ u32x4 uniform_ld16(const void *base, uint32_t bytes, int lane, int byte_offset)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, byte_offset, NT ? (1 << 1) : WHVI_VEC_AUX));
}
// the same load with the low 4 KiB of the byte offset in the instruction's immediate field and only the rest as the scalar
// offset: consecutive chunks then share a scalar offset four at a time and issue back to back (no s_movk between them)
template <bool NT = false>
__device__ __forceinline__ u32x4 uniform_ld16_grouped(const void *base, uint32_t bytes, int lane, int byte_offset)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + (byte_offset & 4095), byte_offset & ~4095,
                                                                            NT ? (1 << 1) : WHVI_VEC_AUX));
}
// the store counterpart; writes beyond `bytes` are dropped.  NT: write-through + non-temporal (see tile_store_stream)
template <bool NT, bool WRITE_THROUGH = true>
__device__ __forceinline__ void uniform_st16(void *base, uint32_t bytes, int lane, int byte_offset, const u32x4 &v)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, lane * 16, byte_offset, NT ? ((WRITE_THROUGH ? (1 << 4) : 0) | (1 << 1)) : 0);
}

// Occupancy target of fwht_rows_kernel.  One-row tiles of 128 data registers (f32 / i32 D = 8192, f64 D = 4096) are
// compiled to 241 VGPRs = two waves per SIMD when left alone.  The f32 one is co-limited by its VALU work (the i32 instance of
// the same kernel, without sign folds, streams at 6.4 TB/s; f32 at 5.6-5.8) and gains from a THIRD wave: 168 VGPRs without
// scratch take (a) its partial tile through bounds-checked buffer accesses and its stores with the chunk offset as the
// scalar offset (no per-chunk address pairs / vector offsets), (b) no tile loop in the code, (c) no SLP register pairs --
// hence its own translation unit, fwht_wide.hip.  5.79 -> 6.07 TB/s (with scratch and the loop: 6.09; profiles/r03/
// rows_store_issue_ab.log).  f64 D = 4096 gains nothing from a third wave (5.88 vs 5.91), i32 needs none.
template <typename T, int K, int ALIGN> constexpr int rows_waves_per_eu()
{
    // (f32 only: f64 gains nothing from the third wave, 5.88 vs 5.91 TB/s)
    constexpr bool wide = K * Elem<T>::VEC * (int)sizeof(typename Elem<T>::acc) / 4 > 64 && std::is_same<T, float>::value;
    return (wide && ALIGN >= 1 && WHVI_WIDE_TILE_WAVES > 0) ? WHVI_WIDE_TILE_WAVES : WHVI_ROWS_WAVES_PER_EU;
}

// ---- batched row FWHT ------------------------------------------------------------------------
// dst/src: n_chunks 16-byte chunks; tile t = chunks [t*64*K, (t+1)*64*K).  Only the last tile
// can be partial; its missing chunks belong to rows that do not exist (rows never straddle
// tiles), so they are read as zero and never stored.
template <typename T, int LOG2D, int K, int POLICY, bool PREFETCH, bool NT, int BLOCK = 256, int ALIGN = 0, bool SIGNED = false>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(rows_waves_per_eu<T, K, ALIGN>())))
fwht_rows_kernel(u32x4 *dst, const u32x4 *src, int64_t n_chunks, int64_t n_tiles)
{
    using E = Elem<T>;
    using A = typename E::acc;
    constexpr int VEC = E::VEC;
    constexpr int TILE = 64 * K;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wpb = BLOCK / 64;
    const int64_t stride = (int64_t)gridDim.x * wpb;
    // Blocks are dealt round-robin over the 8 XCDs; for streams, let the blocks that share an XCD
    // (equal blockIdx % 8) walk one contiguous eighth of the buffer (a speed choice only: +1 % with
    // the write-through stores).  Bijective whenever the grid is a multiple of 8, identity otherwise.
    int64_t blk = blockIdx.x;
    if (NT && !PREFETCH && (gridDim.x & 7) == 0) blk = (blk & 7) * (int64_t)(gridDim.x >> 3) + (blk >> 3);
    int64_t t = blk * wpb + wave;
    // ALIGN (tuning only, one-tile-per-wave launches): 1 = block barrier before the stores, 2 = additionally a
    // barrier after the loads have landed -- experiments on keeping a block's 256 KiB write-back together.
    if (t >= n_tiles) {
        if constexpr (ALIGN >= 2) __syncthreads();
        if constexpr (ALIGN >= 1) __syncthreads();
        return;
    }
    // Store-barrier launches are one tile per wave by construction (their block barriers would not survive a loop whose trip
    // count differs between the waves of a block).  Whether the CODE still contains the grid-stride loop decides nothing
    // functionally and 9 % in time, through ONE thing (tools/probe_rows_forms.py, profiles/r03/rows_loop_form_ab.log,
    // rows_store_issue_ab.log): how the 16 stores of a wave are ISSUED.  With the loop the compiler keeps one vector offset per
    // chunk and computes it right in front of its store (v_or, store, v_or, store, ...); without it, it folds the chunk
    // offsets into immediates and the stores go out back to back -- 6.45 -> 5.89 TB/s on the headline.  The same stores
    // with a single s_nop between them: 6.44 (loop or no loop); with the chunk offset as the scalar offset and no spacing:
    // 5.89, spaced: 6.42.  Loads want the opposite (an s_nop behind each: 6.21).  So: the 4- and 8-byte kernels keep the
    // loop form; 16-bit storage (8 stores per wave, LDS-staged network) is compiled without it: fp16 / bf16 D = 8192
    // 5.5 / 5.3 -> 6.1-6.2 / 6.24-6.36, D <= 4096 +0.5 %.
    // Per-type table of the loop-less form WITH an explicit s_nop between the stores (rows_store_issue_ab.log): f64 D <= 2048
    // +1 % (6.32 -> 6.39), the 128-register f32 / i32 tiles +3 % / +1 % (5.60 -> 5.79, 6.36 -> 6.43); f32 / i32 D <= 4096 and
    // f64 D = 4096 lose 0-4 % and keep the loop.
    constexpr bool WIDE4 = K * VEC * (int)sizeof(A) / 4 > 64 && sizeof(T) == 4;
    constexpr bool AUTO_SINGLE = sizeof(T) == 2 || (WHVI_ALIGN_SINGLE_PASS == -1 && ((sizeof(T) == 8 && LOG2D <= 11) || WIDE4));
    constexpr bool SINGLE_PASS = ALIGN >= 1 && (WHVI_ALIGN_SINGLE_PASS == 1 || (WHVI_ALIGN_SINGLE_PASS < 0 && AUTO_SINGLE));
    // ... and so does the loop form of the f32 D = 512 .. 2048 streams for the last six of its sixteen stores (runs of
    // [2, 1 x 8, 6] in the shipped code object, tools/shipped_isa.py): with one issue slot between all of them the UNSIGNED
    // network streams at 6.42-6.44 TB/s instead of 6.30-6.31 -- what the signed lane stages were adopted for in round 2,
    // without giving up the sign of zero (profiles/r04/stream_forms_store_spacing_ab.log).  D = 4096: 6.42 either way, unchanged.
    constexpr bool F32_MID = std::is_same<T, float>::value && LOG2D >= 9 && LOG2D <= 11 && NT && ALIGN >= 1;
    constexpr bool STORE_NOP = (SINGLE_PASS && sizeof(T) != 2) || F32_MID;   // the loop-less code issues its stores back to back
    extern __shared__ __attribute__((aligned(16))) char whvi_smem[];
    auto transform = [&](A (&r)[K][VEC]) {
        if constexpr (POLICY == POLICY_LDS)
            fwht_tile_lds<A, VEC, K, LOG2D>(r, lane, reinterpret_cast<A *>(whvi_smem) + wave * lds_slab_floats<VEC, K>());
        else
        // SIGNED (f32 streams of D = 512 .. 2048, chosen by the dispatch): the signed DPP stages of fwht_tile.hpp plus ONE
        // repair multiply per element at the end, fma(r, sigma, +0): exact for every non-zero value, and an exact
        // cancellation (+0 under either convention) stays +0.  Bit-identical to the unsigned network except that a
        // NEGATIVE zero result (rows made of signed zeros only: -0 + -0) comes out as +0.  Measured 6.42-6.43 vs
        // 6.30-6.33 TB/s at D = 512 / 1024 / 2048 (D = 4096: 6.0-6.4 vs 6.48, so that shape keeps the unsigned
        // network); profiles/r02/plateau_*.
        if constexpr (SIGNED) {
            static_assert(POLICY == POLICY_DPP && !std::is_same<A, int32_t>::value, "signed form: floating-point arithmetic, DPP network");
            fwht_tile<A, VEC, K, LOG2D, POLICY_DPP, WHVI_ROWS_PKMASK, true, 0>(r, lane);
            constexpr int OUT = fwht_sign_out<VEC, LOG2D>(0);
            const A sg = (__builtin_popcount(lane & OUT) & 1) ? (A)-1 : (A)1;
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int c = 0; c < VEC; ++c) r[k][c] = fma_pm(r[k][c], sg, (A)0);
        } else
            fwht_tile<A, VEC, K, LOG2D, POLICY, WHVI_ROWS_PKMASK>(r, lane);      // no explicit packed adds here: see fwht_tile
    };

#if defined(WHVI_TUNING_BUILD) && defined(WHVI_ROWS_BUFFER_IO)
    // A/B: bounds-checked buffer accesses from the tile's wave-uniform base (the fused kernel's form) -- no full / partial
    // branch, no per-chunk addresses
    auto tile_bytes_of = [&](int64_t tile) -> uint32_t {
        const int64_t base = tile * TILE;
        return (uint32_t)((n_chunks - base < TILE ? n_chunks - base : (int64_t)TILE) * 16);
    };
    auto load_tile = [&](int64_t tile, u32x4 (&raw)[K]) {
        const uint32_t bytes = tile_bytes_of(tile);
#pragma unroll
        for (int k = 0; k < K; ++k) raw[k] = uniform_ld16<NT>(src + tile * TILE, bytes, lane, k * 1024);
    };
    auto store_tile = [&](int64_t tile, A (&r)[K][VEC]) {
        const uint32_t bytes = tile_bytes_of(tile);
#pragma unroll
        for (int k = 0; k < K; ++k) uniform_st16<NT>(dst + tile * TILE, bytes, lane, k * 1024, E::pack(r[k]));
    };
#else
    // One-row tiles of 128 data registers: the partial last tile goes through bounds-checked buffer accesses from the
    // tile's wave-uniform base (reads beyond the buffer return zeros, writes are dropped) instead of 32 guarded accesses
    // with a 64-bit compare and an address pair each -- registers the 168-VGPR budget of three waves per SIMD does not have.
    constexpr bool WIDE = K * VEC * (int)sizeof(A) / 4 > 64 && WHVI_WIDE_TILE_WAVES > 0 && std::is_same<T, float>::value;   // the 3-wave f32 tile, see rows_waves_per_eu
    auto load_tile = [&](int64_t tile, u32x4 (&raw)[K]) {
        const int64_t base = tile * TILE;
        const u32x4 *p = src + base + lane;
        if constexpr (WIDE && NT && WHVI_WIDE_TILE_LOADS == 1) {
            const int64_t left = n_chunks - base;
            const uint32_t bytes = (uint32_t)((left < TILE ? left : (int64_t)TILE) * 16);
#pragma unroll
            for (int k = 0; k < K; ++k) raw[k] = uniform_ld16<NT>(src + base, bytes, lane, k * 1024);
        } else if (base + TILE <= n_chunks) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                raw[k] = ld16<NT>(p + k * 64);
                if constexpr (WHVI_ROWS_LOAD_SPACING > 0) asm volatile("s_nop %0" ::"n"(WHVI_ROWS_LOAD_SPACING - 1));   // A/B: issue spacing
            }
        } else if constexpr (WIDE) {
            const uint32_t bytes = (uint32_t)((n_chunks - base) * 16);
#pragma unroll
            for (int k = 0; k < K; ++k) raw[k] = uniform_ld16<NT>(src + base, bytes, lane, k * 1024);
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                u32x4 z = {0u, 0u, 0u, 0u};
                raw[k] = (base + k * 64 + lane < n_chunks) ? ld16<NT>(p + k * 64) : z;
            }
        }
    };
    auto store_tile = [&](int64_t tile, A (&r)[K][VEC]) {
        const int64_t base = tile * TILE;
        u32x4 *q = dst + base + lane;
        if constexpr (WIDE && NT) {
            // full or partial alike: the chunk offset rides in the instruction's scalar offset, so the 32 stores need one
            // VGPR (lane * 16) between them -- as vector offsets (tile_store_stream) they are 31 loop-invariant registers
            // that the compiler hoists out of the tile loop and then spills
            const int64_t left = n_chunks - base;
            const uint32_t bytes = (uint32_t)((left < TILE ? left : (int64_t)TILE) * 16);
#pragma unroll
            for (int k = 0; k < K; ++k) uniform_st16<true>(dst + base, bytes, lane, k * 1024, E::pack(r[k]));
        } else if (base + TILE <= n_chunks) {
            if constexpr (NT && WHVI_ROWS_STORE_FORM == 1) {        // A/B: chunk offset as the instruction's scalar offset
#pragma unroll
                for (int k = 0; k < K; ++k) uniform_st16<true>(dst + base, TILE * 16, lane, k * 1024, E::pack(r[k]));
            } else if constexpr (NT && WHVI_ROWS_STORE_FORM >= 3) {  // A/B: issue spacing between the stores
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if constexpr (WHVI_ROWS_STORE_FORM == 5) uniform_st16<true>(dst + base, TILE * 16, lane, k * 1024, E::pack(r[k]));
                    else tile_store_stream(dst + base, lane, k, E::pack(r[k]), TILE * 16);
                    if constexpr (WHVI_ROWS_STORE_FORM == 4) __builtin_amdgcn_s_sleep(1);
                    else if constexpr (WHVI_ROWS_STORE_FORM == 6) asm volatile("s_nop 0");
                    else asm volatile("s_nop 7");
                }
            } else if constexpr (NT && WHVI_ROWS_STORE_FORM == 2) {  // A/B: descending chunk order
#pragma unroll
                for (int k = K - 1; k >= 0; --k) tile_store_stream(dst + base, lane, k, E::pack(r[k]), TILE * 16);
            } else if constexpr (NT) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    tile_store_stream(dst + base, lane, k, E::pack(r[k]), TILE * 16);
                    if constexpr (STORE_NOP) asm volatile("s_nop 0");
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) st16<NT>(q + k * 64, E::pack(r[k]));
            }
        } else if constexpr (WIDE) {
            const uint32_t bytes = (uint32_t)((n_chunks - base) * 16);
#pragma unroll
            for (int k = 0; k < K; ++k) uniform_st16<NT>(dst + base, bytes, lane, k * 1024, E::pack(r[k]));
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (base + k * 64 + lane < n_chunks) st16<NT>(q + k * 64, E::pack(r[k]));
        }
    };

#endif
    if constexpr (!PREFETCH) {
        // plain grid-stride form: with grid == tiles/waves this is one tile per wave and out; only
        // one tile's worth of registers is ever live (fits a 1024-thread block at 128 VGPRs)
        for (; t < n_tiles; t += SINGLE_PASS ? n_tiles : stride) {
            A r[K][VEC];
            {
                u32x4 raw[K];
                if constexpr (WHVI_ROWS_SETPRIO == 1) __builtin_amdgcn_s_setprio(3);      // A/B: loads issued at high priority
                load_tile(t, raw);
                if constexpr (WHVI_ROWS_SETPRIO == 1) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int k = 0; k < K; ++k) E::unpack(raw[k], r[k]);
            }
            if constexpr (ALIGN >= 2) __syncthreads();
            transform(r);
            if constexpr (ALIGN >= 1) __syncthreads();
            if constexpr (WHVI_ROWS_SETPRIO == 2) __builtin_amdgcn_s_setprio(3);          // A/B: stores issued at high priority
            // (Tried: a block barrier here so the 16 waves store their 256 KiB together.  A copy microbenchmark
            // gains 5 % from it, the real kernel LOSES 8 %: the waves leave the butterflies microseconds apart
            // and the barrier turns that skew into idle time.  profiles/r01/membench_6_store_alignment.log.)
            store_tile(t, r);
        }
    } else {
        // software-pipelined form: tile t+stride is in flight while tile t is butterflied
        u32x4 raw[K];
        load_tile(t, raw);
        for (;;) {
            A r[K][VEC];
#pragma unroll
            for (int k = 0; k < K; ++k) E::unpack(raw[k], r[k]);
            const int64_t tn = t + stride;
            if (tn < n_tiles) load_tile(tn, raw);
            transform(r);
            store_tile(t, r);
            if (tn >= n_tiles) break;
            t = tn;
        }
    }
}

// ---- rows longer than one wave's registers: high-bit pass ---------------------------------------------
// For D = 2^n beyond the single-wave limit the transform factors as H_D = H_{2^(n-LOW)} (x) H_{2^LOW}: the
// row kernel above does index bits [0, LOW) on the 2^LOW-element pieces, then this kernel does HB more bits
// per pass, ascending, on elements 2^b0 apart (b0 >= LOW: whole 16-byte chunks, so adjacent threads still
// touch adjacent chunks).  Same adds in the same order as the one-pass network -> same bits.  This is the role
// of the reference's fwht_batch2_kernel (src/fwht/cuda/fwht_cuda_kernel.cu:35-67, one radix-4 pass per launch
// for log2 D > 14), with up to 4 stages per pass instead of 2.
template <typename T, int HB>
__global__ void __launch_bounds__(256)
fwht_high_kernel(u32x4 *dst, const u32x4 *src, int64_t n_groups, int log2_stride_chunks)
{
    using E = Elem<T>;
    using A = typename E::acc;
    constexpr int VEC = E::VEC;
    constexpr int R = 1 << HB;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n_groups) return;
    const int64_t stride = (int64_t)1 << log2_stride_chunks;
    const int64_t lo = gid & (stride - 1), hi = gid >> log2_stride_chunks;
    const int64_t base = (hi << (log2_stride_chunks + HB)) | lo;
    A v[R][VEC];
#pragma unroll
    for (int m = 0; m < R; ++m) E::unpack(src[base + m * stride], v[m]);
#pragma unroll
    for (int h = 1; h < R; h *= 2)
#pragma unroll
        for (int m = 0; m < R; ++m)
            if ((m & h) == 0) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) bfly(v[m][c], v[m | h][c]);
            }
#pragma unroll
    for (int m = 0; m < R; ++m) dst[base + m * stride] = E::pack(v[m]);
}

// ---- rows of 4 .. 16 wave tiles in ONE pass: a block per row, the bits above the tile through LDS -------------
// A row of D = 2^(LOW + LOG2W) elements is 2^LOG2W contiguous 64-VGPR tiles (LOW = 12 for 32-bit and 16-bit storage,
// 11 for f64); wave w of the block owns tile w, runs the in-register network on index bits [0, LOW) and then the
// block runs bits [LOW, LOW + LOG2W), which are the wave index, on a transposed view: the tile is handed through LDS
// in two halves (8 of its 16 sixteen-byte register units at a time, W x 8 KiB of LDS), thread (w, lane) collects
// unit slice w of the same lane from all W waves, runs the log2(W) butterfly stages across them, puts the results
// back where it found them and every wave reads its own half back.  Same adds in the same (ascending) order as the
// one-wave network and the reference's radix-2 loop -> same bits; one read and one write of the row instead of the
// two or three passes of the piece + high-bit kernels, and 16-bit storage keeps its single rounding up to D = 65536.
// LDS traffic is 4 x the row (2 writes + 2 reads per element) against 2 x through HBM at a sixth of the bandwidth per
// CU: a fifth of the row's HBM time.  Measurements: DESIGN.md 5.1b, profiles/r02/long_rows_*.log.
template <typename T> constexpr int block_tile_k() { return 256 / (Elem<T>::VEC * (int)sizeof(typename Elem<T>::acc)); }

// One half of the exchange: register units [8 HALF, 8 HALF + 8) of every wave's tile (unit q = bytes [16 q, 16 q + 16)
// of the flattened r[K][VEC]).  Compile-time indices throughout: with plain unrolled loops the f16 instantiation kept
// half the tile in scratch.  Two block barriers; on return only the calling wave reads its own LDS region again, so
// the next half (or row) may overwrite it without another barrier.
template <typename A, int VEC, int K, int LOG2W, int HALF>
__device__ __forceinline__ void block_rows_exchange(A (&r)[K][VEC], char *smem, const int wave, const int lane)
{
    constexpr int W = 1 << LOG2W;
    constexpr int HALF_UNITS = 8;
    constexpr int PIECE_BYTES = HALF_UNITS * 16 / W;      // what one thread takes from each wave
    constexpr int PB = PIECE_BYTES >= 16 ? 16 : PIECE_BYTES;   // LDS access width: 16 bytes, 8 for W = 16
    constexpr int NP = PIECE_BYTES / PB;                  // accesses per source wave
    constexpr int EPP = PB / (int)sizeof(A);              // elements per access
    constexpr int PPW = HALF_UNITS * 16 / PB;             // pieces per wave and lane
    static_assert(K * VEC * (int)sizeof(A) == 256 && LOG2W >= 1 && LOG2W <= 4 && EPP >= 1, "64-VGPR tiles, 2..16 waves");
    typedef uint32_t piece_t __attribute__((ext_vector_type(PB / 4)));
    piece_t *lds = reinterpret_cast<piece_t *>(smem);     // [W][PPW][64] pieces
    auto to_words = [](const A (&x)[EPP]) {
        piece_t v;
        static_for<0, EPP>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (sizeof(A) == 8) {
                const uint64_t u = __builtin_bit_cast(uint64_t, x[i]);
                v[2 * i] = (uint32_t)u;
                v[2 * i + 1] = (uint32_t)(u >> 32);
            } else v[i] = __builtin_bit_cast(uint32_t, x[i]);
        });
        return v;
    };
    auto from_words = [](const piece_t &v, A (&x)[EPP]) {
        static_for<0, EPP>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (sizeof(A) == 8) x[i] = __builtin_bit_cast(A, ((uint64_t)v[2 * i + 1] << 32) | v[2 * i]);
            else {
                const uint32_t w32 = v[i];
                x[i] = __builtin_bit_cast(A, w32);
            }
        });
    };
    // own half out
    static_for<0, PPW>([&](auto P) {
        constexpr int p = decltype(P)::value;
        constexpr int f0 = (HALF * HALF_UNITS * 16 + p * PB) / (int)sizeof(A);
        A x[EPP];
        static_for<0, EPP>([&](auto I) { constexpr int i = decltype(I)::value; x[i] = r[(f0 + i) / VEC][(f0 + i) % VEC]; });
        lds[(wave * PPW + p) * 64 + lane] = to_words(x);
    });
    __syncthreads();
    // slice `wave` of every wave's half, the same lane: butterflies across the wave index
    A x[W][NP][EPP];
    static_for<0, W>([&](auto V) {
        constexpr int v = decltype(V)::value;
        static_for<0, NP>([&](auto J) {
            constexpr int j = decltype(J)::value;
            from_words(lds[(v * PPW + wave * NP + j) * 64 + lane], x[v][j]);
        });
    });
    static_for<0, LOG2W>([&](auto S) {
        constexpr int h = 1 << decltype(S)::value;
        static_for<0, W>([&](auto V) {
            constexpr int v = decltype(V)::value;
            if constexpr ((v & h) == 0)
                static_for<0, NP * EPP>([&](auto EI) {
                    constexpr int e = decltype(EI)::value;
                    bfly(x[v][e / EPP][e % EPP], x[v | h][e / EPP][e % EPP]);
                });
        });
    });
    static_for<0, W>([&](auto V) {
        constexpr int v = decltype(V)::value;
        static_for<0, NP>([&](auto J) {
            constexpr int j = decltype(J)::value;
            lds[(v * PPW + wave * NP + j) * 64 + lane] = to_words(x[v][j]);
        });
    });
    __syncthreads();
    // own half back
    static_for<0, PPW>([&](auto P) {
        constexpr int p = decltype(P)::value;
        constexpr int f0 = (HALF * HALF_UNITS * 16 + p * PB) / (int)sizeof(A);
        A y[EPP];
        from_words(lds[(wave * PPW + p) * 64 + lane], y);
        static_for<0, EPP>([&](auto I) { constexpr int i = decltype(I)::value; r[(f0 + i) / VEC][(f0 + i) % VEC] = y[i]; });
    });
}

// PIPE = false: one row per block and out (grid = rows).  PIPE = true (16-wave blocks: ONE block fits a CU, so nothing
// else hides its load / butterfly / store phases -- profiles/r02/block_rows_trace.log: 29 us per row of which 8 us pass
// between a block's last store and its successor's first load): a persistent block walks rows with a grid stride, stores
// each half of the tile as soon as its exchange is done and asks for the same half of the NEXT row right behind those
// stores, into the registers they vacate, so loads and stores are in flight while the other half is exchanged.
template <typename T, int LOG2W, bool NT, bool PIPE>
__global__ void __launch_bounds__(64 << LOG2W)
    // exactly 4 waves per SIMD: 128 VGPRs, 16 waves per CU (two 512-thread blocks; LDS allows no more than that anyway)
    __attribute__((amdgpu_waves_per_eu(4, 4)))
fwht_block_rows_kernel(u32x4 *dst, const u32x4 *src, int64_t n_rows, uint64_t *trace)   // trace: -DWHVI_BLOCK_TRACE builds only
{
    using E = Elem<T>;
    using A = typename E::acc;
    constexpr int VEC = E::VEC;
    constexpr int K = block_tile_k<T>();
    constexpr int W = 1 << LOG2W;
    constexpr int TILE = 64 * K;                          // storage chunks per wave tile
    constexpr int LOW = ilog2(TILE * VEC);
    // first half of the tile stored as soon as its exchange is done (its write-back drains under the second exchange):
    // always when pipelined; otherwise where the block is bound by its own phases rather than by HBM -- 16 waves (f64
    // 4.85 -> 5.01 TB/s, fp16 4.01 -> 4.2) and 16-bit storage (8 waves 5.66 -> 5.8) -- but not for 8-wave blocks of 4- / 8-byte
    // storage, which lose 1.5 % to it (profiles/r02/block_rows_early_store_ab.log)
    constexpr bool EARLY = PIPE || LOG2W == 4 || sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(16))) char whvi_smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row = blockIdx.x;
    if (NT && !PIPE && (gridDim.x & 7) == 0) row = (row & 7) * (int64_t)(gridDim.x >> 3) + (row >> 3);   // XCD-contiguous
    // trace (tools/probe_block_trace.py, -DWHVI_BLOCK_TRACE builds only): five 100 MHz timestamps + the hardware id per
    // row, written by lane 0 of wave 0.  The scheduling fences stay in production builds: they keep the compiler from
    // hoisting the next phase's loads over this one, which is what holds the kernel at 128 VGPRs without scratch.
    auto stamp = [&](int slot) {
        __builtin_amdgcn_sched_barrier(0);
#ifdef WHVI_BLOCK_TRACE
        if (trace != nullptr && wave == 0 && lane == 0) trace[row * 8 + slot] = wall_clock64();
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    if (row >= n_rows) return;
    auto load_chunks = [&](int64_t rw, u32x4 (&raw)[K], auto HALF) {
        const u32x4 *p = src + (rw * W + wave) * TILE + lane;
        static_for<0, K / 2>([&](auto KK) {
            constexpr int k = decltype(HALF)::value * (K / 2) + decltype(KK)::value;
            raw[k] = ld16<NT>(p + k * 64);
        });
    };
    auto store_chunks = [&](int64_t rw, A (&r)[K][VEC], auto HALF) {
        u32x4 *tile = dst + (rw * W + wave) * TILE;
        static_for<0, K / 2>([&](auto KK) {
            constexpr int k = decltype(HALF)::value * (K / 2) + decltype(KK)::value;
            if constexpr (NT) tile_store_stream(tile, lane, k, E::pack(r[k]), TILE * 16);
            else tile[k * 64 + lane] = E::pack(r[k]);
            if constexpr (NT && WHVI_STORE_SPACING) asm volatile("s_nop 0");     // stores never back to back (5.1, round 3)
        });
    };
    stamp(0);
#ifdef WHVI_BLOCK_TRACE
    if (trace != nullptr && wave == 0 && lane == 0) {
        uint32_t hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        trace[row * 8 + 5] = ((uint64_t)xcc_id << 32) | hw_id;   // (persistent grids: the block's first row only)
    }
#endif
    u32x4 raw[K];
    load_chunks(row, raw, IC<0>{});
    load_chunks(row, raw, IC<1>{});
    // one row; MORE at compile time (the last row is peeled): a run-time condition around the prefetch would keep the
    // old chunks alive into the merge and double the live registers
    auto one_row = [&](auto MORE) {
        constexpr bool more = decltype(MORE)::value;
        const int64_t next = row + gridDim.x;
        A r[K][VEC];
#pragma unroll
        for (int k = 0; k < K; ++k) E::unpack(raw[k], r[k]);
        stamp(1);
        fwht_tile<A, VEC, K, LOW, POLICY_DPP, 0>(r, lane);
        stamp(2);                                             // own tile loaded and transformed
        block_rows_exchange<A, VEC, K, LOG2W, 0>(r, whvi_smem, wave, lane);
        if constexpr (EARLY) store_chunks(row, r, IC<0>{});
        if constexpr (more) load_chunks(next, raw, IC<0>{});
        block_rows_exchange<A, VEC, K, LOG2W, 1>(r, whvi_smem, wave, lane);
        stamp(3);                                             // exchange done
        if constexpr (!EARLY) store_chunks(row, r, IC<0>{});
        store_chunks(row, r, IC<1>{});
        if constexpr (more) load_chunks(next, raw, IC<1>{});
        stamp(4);                                             // stores issued
    };
    if constexpr (PIPE) {
        while (row + gridDim.x < n_rows) {
            one_row(std::true_type{});
            row += gridDim.x;
            stamp(0);
        }
    }
    one_row(std::false_type{});
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
// Division by a launch-invariant 32-bit divisor as multiply-high + shifts (Granlund & Montgomery):
// row -> (group index, sample index) costs a handful of VALU ops instead of a 64-bit division.
struct FastDiv {
    uint32_t d, m, s1, s2;
    __device__ __forceinline__ uint32_t div(uint32_t n) const
    {
        const uint32_t t = __umulhi(m, n);
        return (t + ((n - t) >> s1)) >> s2;
    }
    __device__ __forceinline__ uint32_t mod(uint32_t n) const { return n - div(n) * d; }
};

inline FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while (l < 32 && ((uint64_t)1 << l) < d) ++l;          // l = ceil(log2 d)
    f.m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l < 1 ? 0 : l - 1;
    return f;
}

// Same tile ownership as fwht_rows_kernel (one tile per wave, no loop).  AXIS_COL: scale vectors
// are indexed by the column (idx mod D) and fetched as 16-byte chunks in the same lane layout as
// the data (L1/L2 hits: a and c are D elements shared by every row, b is n_samples*D).  AXIS_ROW:
// one scalar per row.  EYE: src is not read; row i of each group is c[i] * e_i (the first group_rows
// rows of torch.diag(s2), src/weights.py:73).  Every multiply is its own rounding (built with -ffp-contract=off), like
// the reference's separate matmul_diag_left kernels (src/utils.py:4-12).
// (WHVI_FUSED_PKMASK = 2, tuning.hpp: packed adds in the permlane stages only; the TU is built with -fno-slp-vectorize)
//
// STAGE (AXIS_COL): which scale vectors the BLOCK copies into LDS once, to be multiplied straight out of LDS by its waves:
//   STAGE_NONE  every vector is fetched from L2 by the wave that needs it (small launches; per-sample a / c when the rows
//               of a block belong to different samples);
//   STAGE_AC    a and c.  Shared a / c: always possible.  Per-sample a / c: only when all rows of a block belong to ONE
//               sample (rows in (sample, batch, D) order with sample_stride a multiple of the rows per block -- the host
//               checks), the block then stages that sample's vectors;
//   STAGE_ABC   a, b and c of the block's one sample (same condition): no scale vector comes through L2 -> L1 at all.
constexpr int STAGE_NONE = 0, STAGE_AC = 1, STAGE_ABC = 3;

// SHARED_SRC (WHVI_FUSED_SRC_SHARED; AXIS_COL, rows of >= 64 chunks): src holds the rows of ONE sample -- sample_stride of
// them -- shared by all samples, row r reads src row r mod sample_stride.  The (batch, D) input of a layer's first
// Monte-Carlo pass is then read from the caches instead of being expanded to (S, batch, D) in HBM first.
// ONE (WHVI_FUSED_ONE_TRANSFORM; c must be NULL): dst = a (.) FWHT(b_s (.) src) -- the second half of the pipeline alone.
// With a shared source the first half, FWHT(c (.) x), is the same for every sample: it is computed ONCE (a ONE launch with
// b := c) and every sample then costs one transform instead of two -- the same multiplies and butterflies, the same bits.
template <typename T, int LOG2D, int K, int AXIS, bool EYE, bool NT, int BLOCK, int POLICY = POLICY_DPP,
          int STAGE = STAGE_NONE, bool SHARED_SRC = false, bool ONE = false>
__global__ void __launch_bounds__(BLOCK)
fused_shs_kernel(u32x4 *dst, const u32x4 *src, const T *a, const T *b, const T *c,
                 int64_t n_chunks, int64_t n_tiles, FastDiv by_sample_stride, FastDiv by_n_samples,
                 FastDiv by_group_rows, int flags, uint32_t same_sample_blocks)
{
    const bool a_per_sample = flags & WHVI_FUSED_A_PER_SAMPLE;
    const bool c_per_sample = flags & WHVI_FUSED_C_PER_SAMPLE;
    using E = Elem<T>;
    using A = typename E::acc;
    constexpr int VEC = E::VEC;
    constexpr int LV = ilog2(VEC);
    constexpr int TILE = 64 * K;
    constexpr int SH = LOG2D - LV;           // log2(chunks per row)
    constexpr uint32_t CPR = 1u << SH;
    static_assert(LOG2D >= LV, "fused kernel handles rows of at least one chunk");
    static_assert(STAGE == STAGE_NONE || (AXIS == WHVI_AXIS_COL && !EYE && sizeof(A) == sizeof(T)), "staging: f32 / f64 column scales");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // streams: XCD-contiguous block order and a block barrier before the stores, as in fwht_rows_kernel
    int64_t blk = blockIdx.x;
    if (SHARED_SRC && same_sample_blocks == 2u) {
        // shared source, sample index FASTEST within an XCD (host: whole blocks per sample, blocks per sample a multiple of
        // 8): blocks are dealt round-robin over the 8 XCDs; XCD x takes row groups x, x + 8, ... and runs each of them for
        // all samples back to back, so a source tile is fetched into that XCD's L2 once and hit n_samples - 1 times
        // (sample-major order re-reads the whole source per sample from the Infinity Cache: as many fabric bytes in as out)
        const uint32_t b = blockIdx.x, xcd = b & 7u, i = b >> 3;
        const uint32_t q = by_n_samples.div(i), smp = i - q * by_n_samples.d;
        blk = (int64_t)smp * (gridDim.x / by_n_samples.d) + (q * 8u + xcd);
    } else
    if (NT && (gridDim.x & 7) == 0) blk = (blk & 7) * (int64_t)(gridDim.x >> 3) + (blk >> 3);
    int64_t t = blk * (BLOCK / 64) + wave;
    // same_sample_blocks (host: rows in (batch, sample, D) order, one row per tile, whole groups of 4 x n_samples rows,
    // per-sample a / c): the block's waves take rows s, s + S, s + 2 S, s + 3 S of one group -- ONE sample, so its a and c
    // can be staged once per block like shared ones instead of three vectors per tile coming through L2 (where at 16 KiB
    // per vector and 64 samples they no longer all stay: +14 % fabric reads, profiles/r03).  Consecutive blocks take
    // consecutive s: a group of S blocks still covers 4 S contiguous rows.
    // (mode 1 ONLY: under mode 2 -- the shared-source order above -- a block keeps four CONSECUTIVE tiles, which the host
    // has checked lie inside one sample; taking this branch there as well put the waves S tiles apart, i.e. in different
    // samples whenever S does not divide the blocks per sample, while wave 0's sample alone was staged: round-3 bug)
    if (same_sample_blocks == 1u) {
        const uint32_t q = by_n_samples.div((uint32_t)blk), smp = (uint32_t)blk - q * by_n_samples.d;
        t = ((int64_t)q * (BLOCK / 64) + wave) * by_n_samples.d + smp;
    }
    const bool active = t < n_tiles;                          // wave-uniform; a block always has at least one active wave
    const int64_t base = t * TILE;
    const uint32_t row0 = (uint32_t)(base >> SH);             // first row of the tile (wave-uniform)

    auto group_index = [&](uint32_t row) __attribute__((always_inline)) -> uint32_t { return by_group_rows.mod(row); };
    auto sample_index = [&](uint32_t row) __attribute__((always_inline)) -> uint32_t {
        return by_n_samples.mod(by_sample_stride.div(row));
    };

    // ---- request order: the block's scale vectors FIRST (L2 hits, back within a microsecond; loads of one wave return in
    // order, so behind the tile they would only arrive with it), the tile right behind them; the vectors are then written
    // to LDS and the block barrier passes while the tile's loads are still in flight.  (Round 2 staged, synchronised and
    // only then asked for the tile: one exposed L2 round trip + barrier per block, which at two waves per SIMD -- f64
    // rows of 4096 -- nothing hides: 4.8 -> 5.5 TB/s.)
    extern __shared__ __attribute__((aligned(16))) char whvi_smem[];
    constexpr int NSTAGED = STAGE == STAGE_ABC ? 3 : (STAGE == STAGE_AC ? 2 : 0);
    constexpr int STAGED = NSTAGED * (1 << LOG2D);                    // elements of LDS in front of the slabs
    A *const lds_a = reinterpret_cast<A *>(whvi_smem);
    A *const lds_c = lds_a + (1 << LOG2D);
    A *const lds_b = lds_c + (1 << LOG2D);
    typedef A chunk_t __attribute__((ext_vector_type(VEC)));          // one 16-byte chunk of arithmetic values
    constexpr int STG_ITERS = ((1 << LOG2D) + BLOCK * VEC - 1) / (BLOCK * VEC);      // chunks per thread and vector
    chunk_t stg[NSTAGED > 0 ? NSTAGED : 1][STG_ITERS];
    const T *stg_src[3] = {nullptr, nullptr, nullptr};                // c, a, b (the order they are needed in)
    A *const stg_dst[3] = {lds_c, lds_a, lds_b};
    if constexpr (STAGE != STAGE_NONE) {
        // the block's sample: read only where a vector is per-sample (all rows of the block then share it: host-checked)
        const int64_t t_first = same_sample_blocks == 1u ? t - (int64_t)wave * by_n_samples.d : blk * (BLOCK / 64);   // wave 0's tile
        const uint32_t blk_row0 = (uint32_t)((t_first * TILE) >> SH);
        const size_t s_off = (size_t)sample_index(blk_row0) << LOG2D;
        stg_src[0] = (c == nullptr || ONE) ? nullptr : c + (c_per_sample ? s_off : 0);
        stg_src[1] = a == nullptr ? nullptr : a + (a_per_sample ? s_off : 0);
        stg_src[2] = (STAGE == STAGE_ABC && b != nullptr) ? b + s_off : nullptr;
#pragma unroll
        for (int v = 0; v < NSTAGED; ++v)
            if (stg_src[v] != nullptr) {
#pragma unroll
                for (int j = 0; j < STG_ITERS; ++j) {
                    const int i = (threadIdx.x + j * BLOCK) * VEC;
                    if (i < (1 << LOG2D)) stg[v][j] = *reinterpret_cast<const chunk_t *>(stg_src[v] + i);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the tile: bounds-checked buffer accesses from its (wave-uniform) base -- chunks of a partial last tile beyond the
    // buffer read as zero and are never written, waves past the last tile touch nothing: no branch, no per-chunk address
    const uint32_t tile_bytes = active ? (uint32_t)((n_chunks - base < TILE ? n_chunks - base : (int64_t)TILE) * 16) : 0u;
    u32x4 raw[EYE ? 1 : K];
    if constexpr (!EYE) {
#if defined(WHVI_TUNING_BUILD) && WHVI_FUSED_TILE_LOADS == 2   /* timing experiment only: every tile taken to be full (WRONG on ragged tails) */
        if (active) {
#pragma unroll
            for (int k = 0; k < K; ++k) raw[k] = ld16<NT>(src + base + k * 64 + lane);
        }
        if (false)
#else
        // Full tiles of 4-byte elements (and short f64 rows): plain global loads behind a wave-uniform branch -- on a bare
        // stream buffer loads are 10 % slower than global loads (5.84 vs 6.48 TB/s on the plain transform) and here
        // 1-3.5 % (f32 D = 512 / 2048 / 4096: shared a / c 6.44 / 6.39 / 6.32 vs 6.37 / 6.38 / 6.31 TB/s, three L2 vectors
        // 5.98 / 5.92 / 5.54 vs 5.84 / 5.72 / 5.42); f64 rows of 2048 and 4096 lose with them (6.18 vs 6.25, and 3.8 vs
        // 5.95: the branch costs the 128-register tile its second wave per SIMD) and keep the bounds-checked buffer loads,
        // as do the partial last tile and idle waves everywhere.  gpurun_out r03_ab_{prod,tl1,tl2}.log
        constexpr bool GLOBAL_TILE_LOADS = WHVI_FUSED_TILE_LOADS >= 0 ? WHVI_FUSED_TILE_LOADS == 1 : (sizeof(A) == 4 || LOG2D <= 10);
        if constexpr (SHARED_SRC) {
            static_assert(AXIS == WHVI_AXIS_COL && SH >= 6, "shared source: column axis, rows of >= 64 chunks");
            constexpr int KPR = (int)CPR / 64;                       // k-steps per row
            if (active) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t srow = by_sample_stride.mod(row0 + (uint32_t)(k / KPR));      // wave-uniform
                    raw[k] = ld16<false>(src + (int64_t)srow * CPR + (k % KPR) * 64 + lane);     // cached: meant to be resident
                }
            }
        } else
        if (GLOBAL_TILE_LOADS && tile_bytes == TILE * 16) {
#pragma unroll
            for (int k = 0; k < K; ++k) raw[k] = ld16<NT>(src + base + k * 64 + lane);
        } else
#endif
        {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (WHVI_FUSED_TILE_LOADS_GROUPED == 1 || (WHVI_FUSED_TILE_LOADS_GROUPED < 0 && sizeof(T) == 8))
                    raw[k] = uniform_ld16_grouped<NT>(src + base, tile_bytes, lane, k * 1024);      // four per scalar offset: back to back
                else
                    raw[k] = uniform_ld16<NT>(src + base, tile_bytes, lane, k * 1024);
            }
        }
    }
    if constexpr (STAGE != STAGE_NONE) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < NSTAGED; ++v)
            if (stg_src[v] != nullptr) {
#pragma unroll
                for (int j = 0; j < STG_ITERS; ++j) {
                    const int i = (threadIdx.x + j * BLOCK) * VEC;
                    if (i < (1 << LOG2D)) *reinterpret_cast<chunk_t *>(stg_dst[v] + i) = stg[v][j];
                }
            }
        __syncthreads();
    }
    if (!active) {
        if constexpr (NT) __syncthreads();      // the store-alignment barrier below
        return;
    }

    // The two transforms of the pipeline use the signed DPP form (fwht_tile.hpp): the first leaves the tile with
    // sigma = (-1)^popcount(lane & mask), the scalings between them commute with it, the second takes it back to 0.
    constexpr bool SIGNED = WHVI_FUSED_SIGNED && POLICY == POLICY_DPP && !ONE;   // (one transform alone: the unsigned network)
    constexpr int SIGN_MID = SIGNED ? fwht_sign_out<VEC, LOG2D>(0) : 0;
    static_assert(!SIGNED || fwht_sign_out<VEC, LOG2D>(SIGN_MID) == 0, "two transforms restore the sign convention");
    auto transform = [&](A (&r)[K][VEC], auto second) {
        if constexpr (POLICY == POLICY_LDS)
            fwht_tile_lds<A, VEC, K, LOG2D>(r, lane, reinterpret_cast<A *>(whvi_smem) + STAGED +
                                                         wave * lds_slab_floats<VEC, K>());
        else
            fwht_tile<A, VEC, K, LOG2D, POLICY_DPP, WHVI_FUSED_PKMASK, SIGNED, decltype(second)::value ? SIGN_MID : 0>(r, lane);
    };

    // rows never straddle tiles and TILE is a multiple of CPR or vice versa
    auto chunk_row = [&](int k) __attribute__((always_inline)) -> uint32_t {
        if constexpr (SH >= 6) return row0 + (uint32_t)((k * 64) >> SH);                 // wave-uniform
        else return row0 + (uint32_t)((k * 64 + lane) >> SH);
    };
    auto chunk_col = [&](int k) __attribute__((always_inline)) -> uint32_t { return (uint32_t)(k * 64 + lane) & (CPR - 1); };
    // scale factors of chunk k: VEC column values (AXIS_COL) or one row scalar broadcast.  Rows of >= 64 chunks (UNIFORM):
    // the row's sample -- hence its vector's base -- is wave-uniform, computed ONCE per row of the tile (branch-free: a
    // per-chunk branch on per_sample would cut the phases below into basic blocks the scheduling fences cannot order);
    // chunk k of the lane then sits at a compile-time offset from that base plus lane * 16.
    constexpr bool UNIFORM = AXIS == WHVI_AXIS_COL && SH >= 6;
    constexpr int TILE_ROWS = UNIFORM ? ((K * 64) >> SH) : 1;         // rows of the tile (1, 2, 4, ...)
    constexpr int CHUNKS_PER_ROW_HERE = UNIFORM ? (int)CPR / 64 : 1;   // k-steps per row
    struct RowBases { const T *p[TILE_ROWS]; };
    auto row_bases = [&](const T *vec, bool per_sample) __attribute__((always_inline)) {
        RowBases rb;
        const uint32_t keep = per_sample ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int j = 0; j < TILE_ROWS; ++j)
            rb.p[j] = vec + ((size_t)(sample_index(row0 + (uint32_t)j) & keep) << LOG2D);
        return rb;
    };
    auto scale = [&](const T *vec, bool per_sample, const RowBases &rb, int k, A (&out)[VEC]) {
        if constexpr (UNIFORM) {
            // f64: four loads per scalar offset (the low 3 KiB of the chunk offset as immediates), so the loads of a vector
            // issue back to back instead of one s_movk apart: f64 D = 4096 shared 5.88 -> 6.02 TB/s, per-sample D <= 2048
            // +1 %; f32 +-0.2 % either way (profiles/r03/fused_vector_load_issue_ab.log) and keeps the plain form
            if constexpr (WHVI_VEC_LOAD_GROUPED == 1 || (WHVI_VEC_LOAD_GROUPED < 0 && sizeof(T) == 8))
                E::unpack(uniform_ld16_grouped(rb.p[k / CHUNKS_PER_ROW_HERE], (uint32_t)sizeof(T) << LOG2D, lane,
                                               (k % CHUNKS_PER_ROW_HERE) * 1024), out);
            else
                E::unpack(uniform_ld16(rb.p[k / CHUNKS_PER_ROW_HERE], (uint32_t)sizeof(T) << LOG2D, lane,
                                       (k % CHUNKS_PER_ROW_HERE) * 1024), out);
        } else if constexpr (AXIS == WHVI_AXIS_COL) {
            const uint32_t vec_base = per_sample ? sample_index(chunk_row(k)) << LOG2D : 0u;
            E::unpack(*reinterpret_cast<const u32x4 *>(vec + (size_t)vec_base + chunk_col(k) * VEC), out);
        } else {
            const uint32_t vec_base = per_sample ? sample_index(chunk_row(k)) * by_group_rows.d : 0u;
            const A v = (A)vec[(size_t)vec_base + group_index(chunk_row(k))];
#pragma unroll
            for (int e = 0; e < VEC; ++e) out[e] = v;
        }
    };

    A r[K][VEC];

    // vectors staged in LDS: multiply straight out of LDS, chunk by chunk -- no 64-register copy of a
    // vector is ever live across a transform
    auto apply_staged = [&](const A *staged) __attribute__((always_inline)) {
        // eight 16-byte LDS reads in flight, then their eight multiplies (left alone the compiler reads two chunks at a
        // time and waits for each pair: the LDS latency eight times per vector)
        constexpr int G = K < 8 ? K : 8;
#pragma unroll
        for (int k0 = 0; k0 < K; k0 += G) {
            chunk_t v[G];
#pragma unroll
            for (int g = 0; g < G; ++g) v[g] = *reinterpret_cast<const chunk_t *>(staged + chunk_col(k0 + g) * VEC);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int e = 0; e < VEC; ++e) r[k0 + g][e] = v[g][e] * r[k0 + g][e];
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Scale vectors that are not staged in LDS (the per-sample vector b unless the block has one sample; a / c when they
    // are per-sample in (batch, sample, D) order, or the launch is small) are fetched from L2 where they are used -- AFTER
    // the transform in front of them (requesting a whole vector before the transform keeps 64 more registers live across
    // ~1000 butterfly instructions: 184-202 VGPRs = 2 waves per SIMD).  Round 3: even so, a whole vector in flight next
    // to the tile is 64 + 64 registers plus addressing -- 134-146 VGPRs, THREE waves per SIMD -- so five eighths of
    // the chunks are requested up front and the rest only when as many of the first ones have been multiplied in and
    // their registers are free: 64 + 40 registers, FOUR waves per SIMD, at the price of a second (short, mostly
    // overlapped) wait per vector.  Tiles of 128 data registers (one f64 row of 4096 per wave) do the same to stay within the 256
    // registers that two waves per SIMD allow (round 2: 256 VGPRs + 76 AGPRs, one wave, 2.9 TB/s).
    constexpr bool BIG_TILE = K * VEC * (int)sizeof(A) / 4 > 64;      // one f64 row of 4096 per wave: 128 data registers
    // how much up front, measured per shape (tools/ab_fused_r03.sh, 4 GiB in place, interleaved builds; TB/s for 5/8, 6/8
    // and all of the vector up front): f32 with three L2 vectors D = 512 5.67 / 5.72 / 5.87, D = 2048 5.61 / 5.64 / 5.75,
    // D = 4096 5.42 / 5.44 / 5.22; f64 D = 512 5.53 / 5.59 / 5.42, D = 2048 5.38 / 5.41 / 5.19; one f64 row of 4096 per
    // wave with a / c staged (only b comes from L2) 5.72 / 5.79 / 5.91.
    constexpr int EIGHTHS = WHVI_FUSED_UPFRONT_8THS > 0 ? WHVI_FUSED_UPFRONT_8THS
                          : (BIG_TILE && STAGE != STAGE_NONE) ? 8
                          : (sizeof(A) == 4 && STAGE == STAGE_NONE && LOG2D <= 11) ? 8
                          : (STAGE == STAGE_NONE ? 6 : 5);
    constexpr int UPFRONT = K >= 4 ? (K * EIGHTHS) / 8 : K;
    constexpr int LATE = K - UPFRONT;
    auto scale_chunkwise = [&](const T *vec, bool per_sample) __attribute__((always_inline)) {
        RowBases rb;
        if constexpr (UNIFORM) rb = row_bases(vec, per_sample);
        if constexpr (K >= 4 && AXIS == WHVI_AXIS_COL && LATE > 0) {
            A v[K][VEC];
#pragma unroll
            for (int k = 0; k < UPFRONT; ++k) scale(vec, per_sample, rb, k, v[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < LATE; ++k)
#pragma unroll
                for (int e = 0; e < VEC; ++e) r[k][e] = v[k][e] * r[k][e];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = UPFRONT; k < K; ++k) scale(vec, per_sample, rb, k, v[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = LATE; k < K; ++k)
#pragma unroll
                for (int e = 0; e < VEC; ++e) r[k][e] = v[k][e] * r[k][e];
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                A v[VEC];
                scale(vec, per_sample, rb, k, v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) r[k][e] = v[e] * r[k][e];
            }
        }
    };

    // ---- unpack (or synthesise) + first scale
    if constexpr (EYE) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t i = group_index(chunk_row(k));   // row i of the identity (i < group_rows <= D)
            const uint32_t cbase = c_per_sample ? sample_index(chunk_row(k)) * by_group_rows.d : 0u;
            const A cv = (c != nullptr) ? (A)c[(size_t)cbase + i] : (A)1;
#pragma unroll
            for (int e = 0; e < VEC; ++e) r[k][e] = (chunk_col(k) * VEC + e == i) ? cv : (A)0;
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) E::unpack(raw[k], r[k]);
        if constexpr (!ONE) {
            if (c != nullptr) {
                if constexpr (STAGE != STAGE_NONE) apply_staged(lds_c);
                else scale_chunkwise(c, c_per_sample);
            }
        }
    }
    if constexpr (!ONE) transform(r, IC<0>{});
    if (b != nullptr) {
        if constexpr (STAGE == STAGE_ABC) apply_staged(lds_b);
        else scale_chunkwise(b, true);
    }
    transform(r, IC<1>{});
    if (a != nullptr) {
        if constexpr (STAGE != STAGE_NONE) apply_staged(lds_a);
        else scale_chunkwise(a, a_per_sample);
    }
    if constexpr (NT) __syncthreads();          // the block's 4 waves write their 64 KiB back together
    if constexpr (SHARED_SRC && NT && WHVI_FUSED_SHARED_GLOBAL_STORES) {
        // A/B (measurement builds): the write-dominated launch on a shared, cache-resident source with back-to-back global
        // stores like the write-only weight construction -- it LOSES (one transform per sample: 5.66 -> 5.25 TB/s written at
        // D = 2048, 5.64 -> 5.32 at D = 512; two transforms: +-1 %), the spaced buffer stores below stay
        if (tile_bytes == (uint32_t)(64 * K * 16)) {
            u32x4 *q = dst + base + lane;
#pragma unroll
            for (int k = 0; k < K; ++k) st16<true>(q + k * 64, E::pack(r[k]));
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if constexpr (WHVI_FUSED_STORE_FORM == 1)        // A/B: the chunk offset in the VECTOR offset (one v_or per store)
            uniform_st16<NT, !(SHARED_SRC && WHVI_FUSED_SHARED_PLAIN_NT)>(dst + base, tile_bytes, lane + k * 64, 0, E::pack(r[k]));
        else
            uniform_st16<NT, !(SHARED_SRC && WHVI_FUSED_SHARED_PLAIN_NT)>(dst + base, tile_bytes, lane, k * 1024, E::pack(r[k]));
    }
}

// Rows shorter than one 16-byte chunk (D = 1, 2 for f32; D = 1 for f64): one thread per row, same
// arithmetic order as the tiled kernel.  These shapes are tiny by construction (WHVILinear(2, n),
// WHVILinear(1, 1)); they exist so that every shape the reference accepts works on the GPU too.
template <typename T, int LOG2D, int AXIS, bool EYE>
__global__ void fused_small_kernel(T *dst, const T *src, const T *a, const T *b, const T *c, int64_t rows,
                                   FastDiv by_sample_stride, FastDiv by_n_samples, FastDiv by_group_rows, int flags)
{
    using A = typename Elem<T>::acc;
    constexpr int D = 1 << LOG2D;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint32_t s = by_n_samples.mod(by_sample_stride.div((uint32_t)r));
    const uint32_t i = by_group_rows.mod((uint32_t)r);
    const uint32_t unit = (AXIS == WHVI_AXIS_COL) ? (uint32_t)D : by_group_rows.d;
    auto factor = [&](const T *vec, bool per_sample, int j) -> A {
        const size_t base = per_sample ? (size_t)s * unit : 0;
        return (A)vec[base + (AXIS == WHVI_AXIS_COL ? (uint32_t)j : i)];
    };
    A v[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if constexpr (EYE) {
            const A cv = (c != nullptr) ? factor(c, flags & WHVI_FUSED_C_PER_SAMPLE, j) : (A)1;
            v[j] = ((uint32_t)j == i) ? cv : (A)0;
        } else {
            v[j] = (A)src[r * D + j];
            if (c != nullptr) v[j] = factor(c, flags & WHVI_FUSED_C_PER_SAMPLE, j) * v[j];
        }
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int h = 1; h < D; h *= 2)
#pragma unroll
            for (int j = 0; j < D; ++j)
                if ((j & h) == 0) bfly(v[j], v[j | h]);
        const T *vec = pass == 0 ? b : a;
        if (vec != nullptr) {
#pragma unroll
            for (int j = 0; j < D; ++j)
                v[j] = factor(vec, pass == 0 ? true : (bool)(flags & WHVI_FUSED_A_PER_SAMPLE), j) * v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) dst[r * D + j] = (T)v[j];
}

}  // namespace whvi

