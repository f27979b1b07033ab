#pragma once
// whvi_amd/csrc/wbar_fwd.hpp -- the as-written weight construction of src/weights.py:73,
//     W[j,k] = S1_j . fwht( diag(u[j,k]) . fwht( diag(s2_j) ) )  (+ base[j]),
// as its own kernel: zero HBM reads for the construction, one streaming write, and -- with `base` -- the
// `w_bar(g_mu) + w_bar(g_sigma * eps_k)` sum of src/weights.py:93 folded into the epilogue (the mean matrix
// is built once by a first launch and re-read from L2 / Infinity Cache by every sample).
//
// Same bits as the generic fused kernel with the identity input (fused_shs_kernel<EYE>): row i of diag(s2) is
// one-hot, the butterflies of a one-hot row are exact, so fwht(diag(s2))[i,:] = s2_i * H[i,:] is generated
// from the parity of popcount(i & d) instead of being transformed; u_i * (+/- s2_i) = +/- (u_i * s2_i) rounds
// identically.  One transform per row instead of two.
#include "kernels.hpp"

namespace whvi {

// rows are (J, S, R) x D; s1 / s2 are (J, D); matrix (j, k) reads row j * u_group + u_first + k of u (so a
// (J, 1 + S, D) buffer serves both the mean call -- u_first = 0, S = 1 -- and the per-sample call -- u_first = 1);
// base is (J, R, D) or null
// INLINE_MEAN (round 3, cache-resident sizes): W[j,k] = w_bar(u[j,0]) + w_bar(u[j,1+k]) with the MEAN term computed by the
// same wave -- one more in-register transform per tile -- instead of being built by a launch of its own and re-read by
// every sample.  Same multiplies, same butterflies, same final add (mean + sample) as the two-launch form: same bits.
template <typename T, int LOG2D, int K, bool NT, bool INLINE_MEAN = false>
__global__ void __launch_bounds__(256)
wbar_fwd_kernel(u32x4 *dst, const T *s1, const T *u, const T *s2, const u32x4 *base, int64_t n_chunks,
                int64_t n_tiles, uint32_t n_rows, FastDiv by_r, FastDiv by_s, uint32_t u_group, uint32_t u_first,
                uint32_t xcd_blocks, uint32_t n_mats)
{
    using E = Elem<T>;
    using A = typename E::acc;
    static_assert(sizeof(A) == sizeof(T), "f32 / f64 only");
    constexpr int VEC = E::VEC;
    constexpr int LV = ilog2(VEC);
    constexpr int TILE = 64 * K;
    constexpr int SH = LOG2D - LV;
    constexpr uint32_t CPR = 1u << SH;
    constexpr uint32_t D = 1u << LOG2D;
    static_assert(LOG2D >= LV, "rows of at least one chunk");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t blk = blockIdx.x;
    if (xcd_blocks > 0) {
        // with a mean matrix (`base`) every sample's matrix re-reads the same (J, R, D) values: give each XCD (blocks
        // are dealt round-robin over the 8 XCDs) the SAME eighth of every matrix, so its 4 MiB L2 serves that eighth
        // of `base` to all samples instead of each XCD pulling the whole matrix out of the Infinity Cache per sample.
        // Within an XCD the MATRIX index runs fastest: consecutive blocks add the same 64 KiB of `base` to every sample,
        // so the reuse distance is a few blocks whatever the matrix size (sample-major order re-reads an eighth of the
        // matrix per sample: 8 MiB at D = 4096, twice the L2).
        // xcd_blocks = blocks per matrix / 8 (host: only when that divides evenly; a bijection on the block index)
        const uint32_t xcd = (uint32_t)blk & 7u, q = (uint32_t)(blk >> 3);
        const uint32_t w = q / n_mats, m = q - w * n_mats;
        blk = ((int64_t)m * 8 + xcd) * (int64_t)xcd_blocks + w;
    } else if (NT && (gridDim.x & 7) == 0) {
        blk = (blk & 7) * (int64_t)(gridDim.x >> 3) + (blk >> 3);   // XCD-contiguous
    }
    const int64_t t = blk * 4 + wave;
    if (t >= n_tiles) {
        if constexpr (NT) __syncthreads();
        return;
    }
    const int64_t tile0 = t * TILE;
    const bool full = tile0 + TILE <= n_chunks;
    const uint32_t row0 = (uint32_t)(tile0 >> SH);
    auto chunk_row = [&](int k) -> uint32_t {
        if constexpr (SH >= 6) return row0 + (uint32_t)((k * 64) >> SH);
        else return row0 + (uint32_t)((k * 64 + lane) >> SH);
    };
    auto chunk_col = [&](int k) -> uint32_t { return (uint32_t)(k * 64 + lane) & (CPR - 1); };

    A r[K][VEC];
    A mean[INLINE_MEAN ? K : 1][VEC];
    A s1v[K];
    uint32_t base_row[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t row = chunk_row(k) < n_rows ? chunk_row(k) : 0u;   // rows past the end: valid operands, never stored
        const uint32_t jk = by_r.div(row);
        const uint32_t i = row - jk * by_r.d;
        const uint32_t j = by_s.div(jk);
        const uint32_t urow = j * u_group + u_first + (jk - j * by_s.d);
        const A s2v = (A)s2[(size_t)j * D + i];
        const A v = (A)u[(size_t)urow * D + i] * s2v;
        s1v[k] = (A)s1[(size_t)j * D + i];
        base_row[k] = j * by_r.d + i;
        // H[i,d] = (-1)^popcount(i & d), d = d0 + e with d0 a multiple of VEC: one parity per chunk, one per position
        const bool par_k = __builtin_popcount(i & (chunk_col(k) * VEC)) & 1;
#pragma unroll
        for (int e = 0; e < VEC; ++e) r[k][e] = (par_k != (bool)(__builtin_popcount(i & (uint32_t)e) & 1)) ? -v : v;
        if constexpr (INLINE_MEAN) {
            const A v0 = (A)u[(size_t)(j * u_group) * D + i] * s2v;       // row 0 of the group: the mean vector
#pragma unroll
            for (int e = 0; e < VEC; ++e) mean[k][e] = (par_k != (bool)(__builtin_popcount(i & (uint32_t)e) & 1)) ? -v0 : v0;
        }
    }
    if constexpr (INLINE_MEAN) {
        fwht_tile<A, VEC, K, LOG2D, POLICY_DPP, 0>(mean, lane);
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) mean[k][e] = s1v[k] * mean[k][e];
    }
    fwht_tile<A, VEC, K, LOG2D, POLICY_DPP, 0>(r, lane);
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) r[k][e] = s1v[k] * r[k][e];
        if constexpr (INLINE_MEAN) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) r[k][e] = mean[k][e] + r[k][e];
        }
        if (base != nullptr) {
            A m[VEC];
            E::unpack(base[(size_t)base_row[k] * CPR + chunk_col(k)], m);
#pragma unroll
            for (int e = 0; e < VEC; ++e) r[k][e] = m[e] + r[k][e];
        }
    }
    if constexpr (NT) __syncthreads();
    // A WRITE-ONLY stream: plain non-temporal global stores.  The write-through (sc1) buffer stores that win on the read +
    // write streams of the transform kernels (+3 % there) LOSE here: 6.35 vs 6.80 TB/s at D = 2048 x 256 (4 GiB), 6.41 vs
    // 6.85 at D = 512 x 2048, 6.92 vs 7.25 at D = 4096 x 32; a tie at 1 GiB (6.60 vs 6.53) -- measurement builds
    // -DWHVI_WBAR_FWD_STORE=0|1 (0: sc1 nt buffer stores, 1: cached stores), tools/probe_wbar_fwd_stream.py,
    // profiles/r03/write_stream_store_form_ab.log.  torch's fill on the same 4 GiB: 6.93 TB/s.
#if defined(WHVI_TUNING_BUILD) && defined(WHVI_WBAR_FWD_STORE)
    constexpr int STORE_FORM = WHVI_WBAR_FWD_STORE;
#else
    constexpr int STORE_FORM = 2;
#endif
    if (NT && full && STORE_FORM == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) tile_store_stream(dst + tile0, lane, k, E::pack(r[k]), TILE * 16);
    } else if (NT && full) {
#pragma unroll
        // (back to back: an s_nop between the stores -- which a read + write stream wants, DESIGN.md 5.1 round 3 -- costs a
        // write-only stream 6-9 %: 6.24 vs 6.84 TB/s at D = 2048 x 256, 6.74 vs 7.27 at D = 4096 x 32)
        for (int k = 0; k < K; ++k) st16<(STORE_FORM == 2)>(dst + tile0 + k * 64 + lane, E::pack(r[k]));
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (full || tile0 + k * 64 + lane < n_chunks) st16<false>(dst + tile0 + k * 64 + lane, E::pack(r[k]));
    }
}

template <typename T, int LOG2D>
inline void launch_wbar_fwd(void *dst, const void *s1, const void *u, const void *s2, const void *base, int64_t rows,
                            int64_t S, int64_t R, int64_t u_group, int64_t u_first, bool inline_mean, hipStream_t st)
{
    constexpr int K = pick_k<T, LOG2D>();
    constexpr int VEC = Elem<T>::VEC;
    const int64_t n_chunks = (rows << LOG2D) / VEC;
    const int64_t n_tiles = (n_chunks + 64 * K - 1) / (64 * K);
    const FastDiv dr = make_fastdiv((uint32_t)R), ds = make_fastdiv((uint32_t)S);
    const unsigned grid = (unsigned)((n_tiles + 3) / 4);
    // Cache-resident results: quarter-size tiles (never less than one row), four times the waves with a quarter of the
    // work each -- same reasoning and threshold as the backward kernel (wbar_bwd.hpp); tuning builds:
    // WHVI_WBAR_FWD_TILES=big|small overrides (read once).
    constexpr int LVc = ilog2(VEC);
    constexpr int NEED = (LOG2D > LVc + 6) ? (1 << (LOG2D - LVc - 6)) : 1;
    constexpr int KS = NEED > 4 ? NEED : 4;
    if (inline_mean) {
        // the mean term computed in the same launch (whvi_wbar_fwd_mean_*): quarter-size tiles, cached stores.  Two tiles
        // live at once: rows of one 128-register tile (f32 D = 8192, f64 D = 4096) are refused by the dispatch
        constexpr int KM = KS < K ? KS : K;
        // the dispatch admits this form exactly where two 64-register tiles fit (log2d <= multi_pass_low_log2d); tie the two
        // statements of that rule together, so that a shape the dispatch lets through can never fall out of this branch
        // with nothing launched and WHVI_OK returned
        static_assert(tile_vgprs<T, KM>() <= 64 || LOG2D > multi_pass_low_log2d<T>(),
                      "whvi_wbar_fwd_mean: the dispatch bound and the two-tile register budget disagree");
        if constexpr (tile_vgprs<T, KM>() <= 64) {
            const int64_t tiles_m = (n_chunks + 64 * KM - 1) / (64 * KM);
            note_launch<T>("wbar_fwd_kernel", LOG2D, KM, false, true);
            hipLaunchKernelGGL((wbar_fwd_kernel<T, LOG2D, KM, false, true>), dim3((unsigned)((tiles_m + 3) / 4)), dim3(256), 0, st,
                               (u32x4 *)dst, (const T *)s1, (const T *)u, (const T *)s2, (const u32x4 *)nullptr, n_chunks, tiles_m,
                               (uint32_t)rows, dr, ds, (uint32_t)u_group, (uint32_t)u_first, 0u, 0u);
        }
        return;
    }
    // blocks per matrix / 8 when the XCD-sliced order applies (see the kernel): a mean matrix is added, every matrix
    // is a whole number of 8 x 4-tile groups, and the grid is exactly the matrices' blocks
    static const bool xcd_off = [] { const char *e = WHVI_TUNE_ENV("WHVI_WBAR_FWD_XCD"); return e != nullptr && e[0] == '0'; }();   // A/B switch (tuning builds)
    const uint32_t n_mats = (uint32_t)(R > 0 ? rows / R : 0);
    // (streaming launches only: 1 GiB of D = 2048 matrices 282 -> 194 us, 2 GiB 548 -> 323 = the rate without a mean
    // matrix, D = 4096 x 16 297 -> 254; the cache-resident quarter-tile launches are faster in plain order -- 256 MiB:
    // 48.6 vs 64.2 us; profiles/r02/wbar_fwd_tiles_and_order.log)
    auto xcd_blocks_for = [&](int k, bool streaming) -> uint32_t {
        const int64_t per_matrix = (R << LOG2D) / VEC, group = (int64_t)64 * k * 4 * 8;
        if (!streaming || base == nullptr || xcd_off || per_matrix % group != 0 || n_chunks % per_matrix != 0) return 0u;
        return (uint32_t)(per_matrix / group);
    };
    if constexpr (KS < K) {
        static const char *tune = WHVI_TUNE_ENV("WHVI_WBAR_FWD_TILES");
        const bool small = tune ? tune[0] == 's' : (n_chunks * 16 <= NT_MIN_BYTES);
        if (small) {
            const int64_t tiles_s = (n_chunks + 64 * KS - 1) / (64 * KS);
#ifdef WHVI_TUNING_BUILD
            const bool small_nt = tune && tune[0] == 's' && tune[1] == 'n' && n_chunks * 16 >= NT_MIN_BYTES;     // "sn": experiment
            if (small_nt) {
                note_launch<T>("wbar_fwd_kernel", LOG2D, KS, true);
                hipLaunchKernelGGL((wbar_fwd_kernel<T, LOG2D, KS, true>), dim3((unsigned)((tiles_s + 3) / 4)), dim3(256), 0, st,
                                   (u32x4 *)dst, (const T *)s1, (const T *)u, (const T *)s2, (const u32x4 *)base, n_chunks,
                                   tiles_s, (uint32_t)rows, dr, ds, (uint32_t)u_group, (uint32_t)u_first, xcd_blocks_for(KS, true), n_mats);
                return;
            }
#endif
            note_launch<T>("wbar_fwd_kernel", LOG2D, KS, false);
            hipLaunchKernelGGL((wbar_fwd_kernel<T, LOG2D, KS, false>), dim3((unsigned)((tiles_s + 3) / 4)), dim3(256), 0, st,
                               (u32x4 *)dst, (const T *)s1, (const T *)u, (const T *)s2, (const u32x4 *)base, n_chunks,
                               tiles_s, (uint32_t)rows, dr, ds, (uint32_t)u_group, (uint32_t)u_first, xcd_blocks_for(KS, false), n_mats);
            return;
        }
    }
#define WHVI_FWD(NT)                                                                                        \
    do {                                                                                                    \
        note_launch<T>("wbar_fwd_kernel", LOG2D, K, (bool)NT);                                              \
        hipLaunchKernelGGL((wbar_fwd_kernel<T, LOG2D, K, NT>), dim3(grid), dim3(256), 0, st, (u32x4 *)dst,   \
                           (const T *)s1, (const T *)u, (const T *)s2, (const u32x4 *)base, n_chunks, n_tiles, \
                           (uint32_t)rows, dr, ds, (uint32_t)u_group, (uint32_t)u_first, xcd_blocks_for(K, NT), n_mats); \
    } while (0)
    if (n_chunks * 16 >= NT_MIN_BYTES) WHVI_FWD(true);
    else WHVI_FWD(false);
#undef WHVI_FWD
}

template <typename T>
inline int wbar_fwd_dispatch(void *dst, const void *s1, const void *u, const void *s2, const void *base, int64_t J,
                             int64_t S, int64_t R, int32_t log2d, int64_t u_group, int64_t u_first, void *stream,
                             bool inline_mean = false)
{
    constexpr int LV = ilog2(Elem<T>::VEC);
    g_err[0] = 0;
    if (J < 0 || S < 0 || R < 0) return fail(WHVI_ERR_ARG, "whvi_wbar_fwd: negative size%s", "");
    if (log2d < LV || log2d > max_single_pass_log2d<T>())
        return fail(WHVI_ERR_SIZE, "whvi_wbar_fwd: log2(D)%s = %lld is outside the supported range [%lld, ...]", "",
                    log2d, LV);
    if (inline_mean && log2d > multi_pass_low_log2d<T>())
        return fail(WHVI_ERR_SIZE, "whvi_wbar_fwd_mean: log2(D)%s = %lld is beyond the one-launch form (two 64-register tiles); "
                    "build the mean matrix with whvi_wbar_fwd and pass it as `base`", "", log2d);
    if (R > ((int64_t)1 << log2d)) return fail(WHVI_ERR_ARG, "whvi_wbar_fwd: R%s = %lld exceeds D", "", R);
    if (u_first < 0 || u_group < u_first + S || u_group >= ((int64_t)1 << 31))
        return fail(WHVI_ERR_ARG, "whvi_wbar_fwd: u_group%s = %lld does not hold rows u_first .. u_first + S", "", u_group);
    const int64_t rows = J * S * R;
    if (rows == 0) return WHVI_OK;
    if (rows >= ((int64_t)1 << 32)) return fail(WHVI_ERR_SIZE, "whvi_wbar_fwd: rows are indexed with 32 bits%s", "");
    if (!dst || !s1 || !u || !s2) return fail(WHVI_ERR_ARG, "whvi_wbar_fwd: null pointer%s", "");
    if (((uintptr_t)dst & 15) || ((uintptr_t)base & 15))
        return fail(WHVI_ERR_ALIGN, "whvi_wbar_fwd: %s pointer is not 16-byte aligned", ((uintptr_t)dst & 15) ? "dst" : "base");
    hipStream_t st = (hipStream_t)stream;
#define WHVI_CASE(L)                                                                                       \
    case L:                                                                                                \
        if constexpr (L >= LV && L <= max_single_pass_log2d<T>())                                          \
            launch_wbar_fwd<T, L>(dst, s1, u, s2, base, rows, S, R, u_group, u_first, inline_mean, st);                       \
        break;
    switch (log2d) {
        WHVI_CASE(1) WHVI_CASE(2) WHVI_CASE(3) WHVI_CASE(4) WHVI_CASE(5) WHVI_CASE(6) WHVI_CASE(7)
        WHVI_CASE(8) WHVI_CASE(9) WHVI_CASE(10) WHVI_CASE(11) WHVI_CASE(12) WHVI_CASE(13)
    default: break;
    }
#undef WHVI_CASE
    return after_launch("wbar_fwd");
}

}  // namespace whvi
