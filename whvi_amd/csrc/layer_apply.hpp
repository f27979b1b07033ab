#pragma once
// whvi_amd/csrc/layer_apply.hpp -- the two dense products either side of the square layer in a WHVI regression network, for
// ALL Monte-Carlo samples in one launch each, as the HBM-bound streams they are (rocBLAS runs them at 0.56-0.60 of peak):
//
//   small_k_apply_kernel : out[s, b, n] = sum_{c < K} x[b, c] * W[s, n, c] (+ bias[n]) (relu?)        K = 4 or 8
//       `x_padded @ W.T` of WHVIStackedMatrix for a narrow input (src/weights.py:179-180,195-206: WHVILinear(3, 1024) = 256
//       sub-matrices of D = 4 stacked to a (1024, 4) weight per sample).  Write-only: B x K in, S x B x N out.  Every one of
//       the K products is formed (the as-written sub-matrices are diagonal, so K - 1 of them multiply exact zeros -- and turn a
//       non-finite input into NaN exactly like the dense product does); fused multiply-adds in ascending c like a GEMM's
//       accumulation: with one non-zero term per sum the result is that single rounded product whatever the order.
//   row_dot_kernel       : y[s, b] = sum_i relu?(x[s, b, i]) * w[s, i] (+ bias)
//       `F.linear(x, w[None])` of the transposed WHVIColumnMatrix (src/weights.py:239-251: WHVILinear(1024, 1)).  Read-only:
//       S x B x D in, S x B out; a wave owns whole rows, per-lane partial sums in ascending column order, then a butterfly over
//       the 64 lanes.  Floating-point summation order differs from a GEMV's: same tolerance class as any other BLAS.
#include "dispatch.hpp"

#ifndef WHVI_SMALL_K_UNR
#define WHVI_SMALL_K_UNR 4      // rows in flight per thread (4 / 8 / 16 measured: 0.50 / 0.56 / 0.52 ms at config 4 on one box -- within its run-to-run spread)
#endif

namespace whvi {

// Column-owner layout: a thread keeps the K-wide weight rows of its 4 * CPT output columns in registers for the whole slab of
// batch rows it walks (no LDS, no per-chunk index arithmetic); per row it needs x[b, 0:K] -- the same 16 / 32 bytes for every
// lane of the block (a broadcast load) -- and writes one 16-byte chunk per owned column group.  A block covers TPR = min(256, N/4)
// column chunks: for N = 1024 one 4 KiB output row per step, slab after slab a contiguous write-only stream.
template <typename T, int LOG2K, int CPT, bool NT>      // (T = float; named so that whvi_last_kernel prints the real symbol)
__global__ void __launch_bounds__(256)
small_k_apply_kernel(u32x4 *__restrict__ dst, const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                     uint32_t B, uint32_t cpr, uint32_t tpr, uint32_t slab_rows, uint32_t n_slabs, uint32_t relu_out)
{
    constexpr int KIN = 1 << LOG2K, G = KIN / 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    uint32_t lin = blockIdx.x;
    if (NT && (gridDim.x & 7u) == 0u) lin = (lin & 7u) * (gridDim.x >> 3) + (lin >> 3);      // XCD-contiguous
    const uint32_t s = lin / n_slabs, slab = lin - s * n_slabs;
    const uint32_t b0 = slab * slab_rows, b1 = b0 + slab_rows < B ? b0 + slab_rows : B;
    const uint32_t tcol = threadIdx.x % tpr, rg = threadIdx.x / tpr, n_rg = 256u / tpr;
    f4 wr[CPT][4][G], bv[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const uint32_t chunk = c * tpr + tcol;                       // output columns 4 * chunk .. 4 * chunk + 3
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int gi = 0; gi < G; ++gi)
                wr[c][r][gi] = reinterpret_cast<const f4 *>(w)[((size_t)s * cpr * 4 + (size_t)chunk * 4 + r) * G + gi];
        bv[c] = bias != nullptr ? reinterpret_cast<const f4 *>(bias)[chunk] : f4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4 *out = dst + (size_t)s * B * cpr;
    constexpr int UNR = WHVI_SMALL_K_UNR;             // rows (= 16-byte stores per owned column group) in flight per thread
    for (uint32_t b = b0 + rg; b < b1; b += n_rg * UNR) {
        f4 xv[UNR][G];
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const uint32_t row = b + i * n_rg < b1 ? b + i * n_rg : b1 - 1;
#pragma unroll
            for (int gi = 0; gi < G; ++gi) xv[i][gi] = reinterpret_cast<const f4 *>(x)[(size_t)row * G + gi];
        }
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const uint32_t row = b + i * n_rg;
            if (row < b1) {
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    f4 acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float a = 0.0f;                      // a GEMM's +0-initialised accumulator: zero results come out as +0
#pragma unroll
                        for (int gi = 0; gi < G; ++gi)
#pragma unroll
                            for (int e = 0; e < 4; ++e) a = __builtin_fmaf(xv[i][gi][e], wr[c][r][gi][e], a);
                        acc[r] = a;
                    }
                    if (bias != nullptr) acc = acc + bv[c];
                    if (relu_out) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = (acc[e] > 0.0f || acc[e] != acc[e]) ? acc[e] : 0.0f;
                    }
                    st16<NT>(out + (size_t)row * cpr + c * tpr + tcol, __builtin_bit_cast(u32x4, acc));
                }
            }
        }
    }
}

// y[s, b] = sum_i relu?(x[s, b, i]) * w[s, i] (+ bias[0]); rows of one 16-byte chunk up to one 64-register tile (D = 4 .. 4096)
template <typename T, int LOG2D, bool NT>
__global__ void __launch_bounds__(256)
row_dot_kernel(float *__restrict__ y, const u32x4 *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
               int64_t n_chunks, int64_t n_tiles, uint32_t n_rows, FastDiv by_batch, uint32_t relu_in)
{
    constexpr int K = pick_k<float, LOG2D>(), TILE = 64 * K, SH = LOG2D - 2;
    constexpr uint32_t CPR = 1u << SH, D = 1u << LOG2D;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t blk = blockIdx.x;
    if (NT && (gridDim.x & 7) == 0) blk = (blk & 7) * (int64_t)(gridDim.x >> 3) + (blk >> 3);
    const int64_t t = blk * 4 + wave;
    const bool active = t < n_tiles;
    const int64_t tile0 = t * TILE;
    const uint32_t row0 = (uint32_t)(tile0 >> SH);
    const uint32_t brow0 = (uint32_t)((blk * 4 * TILE) >> SH);
    uint32_t brow1 = (uint32_t)(((blk * 4 + 4) * TILE - 1) >> SH);
    if (brow1 >= n_rows) brow1 = n_rows - 1;
    const uint32_t smp0 = by_batch.div(brow0);
    const bool one_sample = by_batch.div(brow1) == smp0;
    u32x4 raw[K];
    if (active) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            raw[k] = u32x4{0u, 0u, 0u, 0u};
            if (tile0 + k * 64 + lane < n_chunks) raw[k] = ld16<NT>(x + tile0 + k * 64 + lane);
        }
    }
    __shared__ __attribute__((aligned(16))) float lds_w[D];
    if (one_sample) {
        for (uint32_t i = threadIdx.x; i < CPR; i += 256)
            reinterpret_cast<f4 *>(lds_w)[i] = reinterpret_cast<const f4 *>(w + ((size_t)smp0 << LOG2D))[i];
        __syncthreads();
    }
    if (!active) return;
    auto chunk_row = [&](int k) __attribute__((always_inline)) -> uint32_t {
        if constexpr (SH >= 6) return row0 + (uint32_t)((k * 64) >> SH);
        else return row0 + (uint32_t)((k * 64 + lane) >> SH);
    };
    float part[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t col = (uint32_t)(k * 64 + lane) & (CPR - 1);
        f4 wv;
        if (one_sample) wv = reinterpret_cast<const f4 *>(lds_w)[col];
        else {
            const uint32_t row = chunk_row(k) < n_rows ? chunk_row(k) : n_rows - 1;
            wv = reinterpret_cast<const f4 *>(w + ((size_t)by_batch.div(row) << LOG2D))[col];
        }
        f4 xv = __builtin_bit_cast(f4, raw[k]);
        if (relu_in) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = (xv[e] > 0.0f || xv[e] != xv[e]) ? xv[e] : 0.0f;
        }
        float a = xv[0] * wv[0];
#pragma unroll
        for (int e = 1; e < 4; ++e) a = __builtin_fmaf(xv[e], wv[e], a);
        part[k] = a;
    }
    const float b0 = bias != nullptr ? bias[0] : 0.0f;
    if constexpr (SH >= 6) {
        constexpr int KPR = (int)CPR / 64;              // k-steps per row
#pragma unroll
        for (int j = 0; j < K / KPR; ++j) {
            float s = part[j * KPR];
#pragma unroll
            for (int k = 1; k < KPR; ++k) s = s + part[j * KPR + k];
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) s = s + __shfl_xor(s, m, 64);
            const uint32_t row = row0 + (uint32_t)j;
            if (lane == 0 && row < n_rows) y[row] = bias != nullptr ? s + b0 : s;
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float s = part[k];
#pragma unroll
            for (int m = 1; m < (int)CPR; m <<= 1) s = s + __shfl_xor(s, m, 64);
            const uint32_t row = chunk_row(k);
            if ((lane & (CPR - 1)) == 0 && row < n_rows) y[row] = bias != nullptr ? s + b0 : s;
        }
    }
}

inline int small_k_apply_dispatch(void *out, const void *x, const void *w, const void *bias, int64_t S, int64_t B, int64_t N,
                                  int32_t log2k, int32_t flags, void *stream)
{
    g_err[0] = 0;
    if (S < 0 || B < 0 || N < 0) return fail(WHVI_ERR_ARG, "whvi_small_k_apply: negative size%s", "");
    if (log2k != 2 && log2k != 3) return fail(WHVI_ERR_SIZE, "whvi_small_k_apply: K%s = 2^%lld input features (4 or 8 only)", "", log2k);
    if (N % 4 != 0) return fail(WHVI_ERR_ARG, "whvi_small_k_apply: N%s = %lld is not a multiple of 4", "", N);
    if (flags & ~WHVI_APPLY_RELU_OUT) return fail(WHVI_ERR_ARG, "whvi_small_k_apply: unknown flags%s 0x%llx", "", flags);
    const int64_t rows = S * B;
    if (rows == 0 || N == 0) return WHVI_OK;
    if (rows >= ((int64_t)1 << 32)) return fail(WHVI_ERR_SIZE, "whvi_small_k_apply: rows are indexed with 32 bits%s", "");
    if (!out || !x || !w) return fail(WHVI_ERR_ARG, "whvi_small_k_apply: null pointer%s", "");
    if (((uintptr_t)out | (uintptr_t)x | (uintptr_t)w | (uintptr_t)bias) & 15)
        return fail(WHVI_ERR_ALIGN, "whvi_small_k_apply: a pointer%s is not 16-byte aligned", "");
    const uint32_t cpr = (uint32_t)(N / 4);
    uint32_t tpr = 1;
    while (tpr < 256 && tpr * 2 <= cpr && cpr % (tpr * 2) == 0) tpr *= 2;        // largest power of two <= 256 dividing cpr
    const uint32_t cpt = cpr / tpr;
    if (cpt > 4) return fail(WHVI_ERR_SIZE, "whvi_small_k_apply: N%s = %lld needs more than 4 column groups per thread", "", N);
    const int64_t n_rg = 256 / tpr;
    int64_t n_slabs = (8 * (int64_t)num_cu() + S - 1) / S;
    const int64_t most = (B + n_rg * 8 - 1) / (n_rg * 8);
    if (n_slabs > most) n_slabs = most;
    if (n_slabs < 1) n_slabs = 1;
    const int64_t slab_rows = (B + n_slabs - 1) / n_slabs;
    n_slabs = (B + slab_rows - 1) / slab_rows;
    if (n_slabs * S >= ((int64_t)1 << 31)) return fail(WHVI_ERR_SIZE, "whvi_small_k_apply: too many blocks%s", "");
    const bool nt = rows * (int64_t)N * 4 > NT_MIN_BYTES;
    const dim3 grid((unsigned)(n_slabs * S));
    hipStream_t st = (hipStream_t)stream;
#define WHVI_SK(L, C, NTV)                                                                                     \
    do {                                                                                                        \
        note_launch<float>("small_k_apply_kernel", L, C, (bool)NTV);                                            \
        hipLaunchKernelGGL((small_k_apply_kernel<float, L, C, NTV>), grid, dim3(256), 0, st, (u32x4 *)out, (const float *)x, \
                           (const float *)w, (const float *)bias, (uint32_t)B, cpr, tpr, (uint32_t)slab_rows, (uint32_t)n_slabs, \
                           (uint32_t)((flags & WHVI_APPLY_RELU_OUT) ? 1 : 0));                                  \
    } while (0)
#define WHVI_SKC(L, C) do { if (nt) WHVI_SK(L, C, true); else WHVI_SK(L, C, false); } while (0)
#define WHVI_SKL(L)                                                                                             \
    do {                                                                                                        \
        if (cpt == 1) WHVI_SKC(L, 1); else if (cpt == 2) WHVI_SKC(L, 2); else if (cpt == 3) WHVI_SKC(L, 3); else WHVI_SKC(L, 4); \
    } while (0)
    if (log2k == 2) WHVI_SKL(2); else WHVI_SKL(3);
#undef WHVI_SKL
#undef WHVI_SKC
#undef WHVI_SK
    return after_launch("small_k_apply");
}

inline int row_dot_dispatch(void *y, const void *x, const void *w, const void *bias, int64_t S, int64_t B, int32_t log2d,
                            int32_t flags, void *stream)
{
    g_err[0] = 0;
    if (S < 0 || B < 0) return fail(WHVI_ERR_ARG, "whvi_row_dot: negative size%s", "");
    if (log2d < 2 || log2d > 12) return fail(WHVI_ERR_SIZE, "whvi_row_dot: log2(D)%s = %lld is outside [2, 12]", "", log2d);
    if (flags & ~WHVI_APPLY_RELU_IN) return fail(WHVI_ERR_ARG, "whvi_row_dot: unknown flags%s 0x%llx", "", flags);
    const int64_t rows = S * B;
    if (rows == 0) return WHVI_OK;
    if (rows >= ((int64_t)1 << 32)) return fail(WHVI_ERR_SIZE, "whvi_row_dot: rows are indexed with 32 bits%s", "");
    if (!y || !x || !w) return fail(WHVI_ERR_ARG, "whvi_row_dot: null pointer%s", "");
    if (((uintptr_t)x | (uintptr_t)w) & 15) return fail(WHVI_ERR_ALIGN, "whvi_row_dot: a pointer%s is not 16-byte aligned", "");
    const FastDiv db = make_fastdiv((uint32_t)B);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t relu_in = (flags & WHVI_APPLY_RELU_IN) ? 1u : 0u;
#define WHVI_RD(L, NTV)                                                                                        \
    do {                                                                                                        \
        constexpr int K_ = pick_k<float, L>();                                                                  \
        const int64_t n_chunks = (rows << L) / 4, n_tiles = (n_chunks + 64 * K_ - 1) / (64 * K_);               \
        note_launch<float>("row_dot_kernel", L, (bool)NTV);                                                     \
        hipLaunchKernelGGL((row_dot_kernel<float, L, NTV>), dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, (float *)y, \
                           (const u32x4 *)x, (const float *)w, (const float *)bias, n_chunks, n_tiles, (uint32_t)rows, db, relu_in); \
    } while (0)
#define WHVI_CASE(L)                                                                                            \
    case L: { const bool nt = (rows << L) * 4 > NT_MIN_BYTES; if (nt) WHVI_RD(L, true); else WHVI_RD(L, false); } break;
    switch (log2d) {
        WHVI_CASE(2) WHVI_CASE(3) WHVI_CASE(4) WHVI_CASE(5) WHVI_CASE(6) WHVI_CASE(7) WHVI_CASE(8) WHVI_CASE(9) WHVI_CASE(10)
        WHVI_CASE(11) WHVI_CASE(12)
    default: break;
    }
#undef WHVI_CASE
#undef WHVI_RD
    return after_launch("row_dot");
}

}  // namespace whvi
