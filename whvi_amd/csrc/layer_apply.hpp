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

namespace whvi {

template <int LOG2K, bool NT>
__global__ void __launch_bounds__(256)
small_k_apply_kernel(u32x4 *__restrict__ dst, const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                     int64_t n_chunks, int64_t n_tiles, uint32_t n_rows, uint32_t cpr, FastDiv by_cpr, FastDiv by_batch, uint32_t relu_out)
{
    constexpr int KIN = 1 << LOG2K;                 // input features (4 or 8)
    constexpr int K = 16, TILE = 64 * K;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t blk = blockIdx.x;
    if (NT && (gridDim.x & 7) == 0) blk = (blk & 7) * (int64_t)(gridDim.x >> 3) + (blk >> 3);
    const int64_t t = blk * 4 + wave;
    const bool active = t < n_tiles;
    const int64_t tile0 = t * TILE;
    const bool full = tile0 + TILE <= n_chunks;
    // the block's first and last output row -> its sample(s); one sample per block: that sample's (N, KIN) weight goes to LDS
    // transposed to [r][c-group][chunk] so that the lanes of a wave read consecutive 16-byte words
    const uint32_t brow0 = by_cpr.div((uint32_t)(blk * 4 * TILE));
    uint32_t brow1 = by_cpr.div((uint32_t)((blk * 4 + 4) * TILE - 1));
    if (brow1 >= n_rows) brow1 = n_rows - 1;
    const uint32_t smp0 = by_batch.div(brow0);
    const bool one_sample = by_batch.div(brow1) == smp0;
    extern __shared__ __attribute__((aligned(16))) float lds[];        // 4 * (KIN / 4) * cpr f4 words  (+ cpr for the bias)
    constexpr int G = KIN / 4;                      // 16-byte groups per weight row
    f4 *lw = reinterpret_cast<f4 *>(lds);
    f4 *lb = lw + (size_t)4 * G * cpr;
    if (one_sample) {
        const f4 *ws = reinterpret_cast<const f4 *>(w) + (size_t)smp0 * cpr * 4 * G;      // rows n = 4 * chunk + r, G words each
        for (uint32_t i = threadIdx.x; i < cpr * 4 * G; i += 256) {
            const uint32_t n = i / G, gidx = i - n * G, chunk = n >> 2, r = n & 3;
            lw[((size_t)r * G + gidx) * cpr + chunk] = ws[i];
        }
        for (uint32_t i = threadIdx.x; i < cpr; i += 256)
            lb[i] = bias != nullptr ? reinterpret_cast<const f4 *>(bias)[i] : f4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    }
    if (!active) {
        if constexpr (NT) __syncthreads();
        return;
    }
    f4 out[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t c = (uint32_t)(tile0 + k * 64 + lane < n_chunks ? tile0 + k * 64 + lane : n_chunks - 1);   // clamp: valid operands, never stored
        const uint32_t row = by_cpr.div(c), chunk = c - row * cpr;
        const uint32_t s = by_batch.div(row), b = row - s * by_batch.d;
        f4 xv[G];
#pragma unroll
        for (int gidx = 0; gidx < G; ++gidx) xv[gidx] = reinterpret_cast<const f4 *>(x)[(size_t)b * G + gidx];
        f4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = 0.0f;
#pragma unroll
            for (int gidx = 0; gidx < G; ++gidx) {
                const f4 wv = one_sample ? lw[((size_t)r * G + gidx) * cpr + chunk]
                                         : reinterpret_cast<const f4 *>(w)[((size_t)s * cpr * 4 + (size_t)chunk * 4 + r) * G + gidx];
#pragma unroll
                for (int e = 0; e < 4; ++e) a = (gidx == 0 && e == 0) ? xv[0][0] * wv[0] : __builtin_fmaf(xv[gidx][e], wv[e], a);
            }
            acc[r] = a;
        }
        if (bias != nullptr) {
            const f4 bv = one_sample ? lb[chunk] : reinterpret_cast<const f4 *>(bias)[chunk];
            acc = acc + bv;
        }
        if (relu_out) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = (acc[e] > 0.0f || acc[e] != acc[e]) ? acc[e] : 0.0f;
        }
        out[k] = acc;
    }
    if constexpr (NT) __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const u32x4 v = __builtin_bit_cast(u32x4, out[k]);
        if (NT && full) st16<true>(dst + tile0 + k * 64 + lane, v);         // write-only stream: back-to-back nt stores (wbar_fwd.hpp)
        else if (full || tile0 + k * 64 + lane < n_chunks) st16<false>(dst + tile0 + k * 64 + lane, v);
    }
}

// y[s, b] = sum_i relu?(x[s, b, i]) * w[s, i] (+ bias[0]); rows of one 16-byte chunk up to one 64-register tile (D = 4 .. 4096)
template <int LOG2D, bool NT>
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
    if (rows >= ((int64_t)1 << 32) || rows * (N / 4) >= ((int64_t)1 << 32))
        return fail(WHVI_ERR_SIZE, "whvi_small_k_apply: output chunks are indexed with 32 bits%s", "");
    if (!out || !x || !w) return fail(WHVI_ERR_ARG, "whvi_small_k_apply: null pointer%s", "");
    if (((uintptr_t)out | (uintptr_t)x | (uintptr_t)w | (uintptr_t)bias) & 15)
        return fail(WHVI_ERR_ALIGN, "whvi_small_k_apply: a pointer%s is not 16-byte aligned", "");
    const uint32_t cpr = (uint32_t)(N / 4);
    const size_t smem = ((size_t)4 * ((1 << log2k) / 4) + 1) * cpr * 16;
    if (smem > 64 * 1024) return fail(WHVI_ERR_SIZE, "whvi_small_k_apply: N%s = %lld does not fit the block's LDS", "", N);
    const int64_t n_chunks = rows * cpr, n_tiles = (n_chunks + 1023) / 1024;
    const FastDiv dc = make_fastdiv(cpr), db = make_fastdiv((uint32_t)B);
    const bool nt = n_chunks * 16 > NT_MIN_BYTES;
    const dim3 grid((unsigned)((n_tiles + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
#define WHVI_SK(L, NTV)                                                                                        \
    do {                                                                                                        \
        note_launch<float>("small_k_apply_kernel", L, (bool)NTV);                                               \
        hipLaunchKernelGGL((small_k_apply_kernel<L, NTV>), grid, dim3(256), smem, st, (u32x4 *)out, (const float *)x, \
                           (const float *)w, (const float *)bias, n_chunks, n_tiles, (uint32_t)rows, cpr, dc, db, \
                           (uint32_t)((flags & WHVI_APPLY_RELU_OUT) ? 1 : 0));                                  \
    } while (0)
    if (log2k == 2) { if (nt) WHVI_SK(2, true); else WHVI_SK(2, false); }
    else { if (nt) WHVI_SK(3, true); else WHVI_SK(3, false); }
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
        hipLaunchKernelGGL((row_dot_kernel<L, NTV>), dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, (float *)y, \
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
