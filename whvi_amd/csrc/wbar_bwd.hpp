#pragma once
// whvi_amd/csrc/wbar_bwd.hpp -- backward of the as-written weight construction
//     W[j,k] = S1_j . fwht( diag(u[j,k]) . fwht( diag(s2_j) ) )            (src/weights.py:73)
// in ONE launch that reads the incoming gradient once and writes three scalars per row.
//
// The reference gets this gradient from autograd over its op chain: matmul_diag_left backward (an
// elementwise product plus a row sum, src/utils.py:4-12) and FWHTFunction.backward = the same FWHT
// (src/fwht/cuda/fwht.py:14-16).  For row i of matrix (j, k), with gw = dL/dW[j,k,i,:]:
//     g1     = fwht(s1_i * gw)                               adjoint of the outer scale and FWHT
//     t1[d]  = fwht(s2_i e_i)[d] = H[i,d] * s2_i              (exact: butterflies of a one-hot row)
//     dL/du[j,k,i]   = sum_d g1[d] * t1[d]
//     dL/ds2[j,i]   += fwht(u_i * g1)[i] = sum_d H[d,i] * (u_i * g1[d])
//     dL/ds1[j,i]   += sum_d gw[d] * fwht(u_i * t1)[d] = gw[i] * (D * (u_i * s2_i))
//       (fwht(u_i s2_i H[i,:]) is exactly D u_i s2_i e_i: every partial sum is a power-of-two multiple or 0)
// The scaling in front of the transform and the final products are separate roundings, as in the separate ATen
// kernels; the two row sums share one signed sum of g1 (a pruned butterfly, see below) -- autograd's own reduction
// order is unspecified as well.
#include "kernels.hpp"

namespace whvi {

// One lane-bit stage of the pruned butterfly: lanes whose bit LB is clear end up with v(lane) + sgn * v(lane ^ (1 << LB))
// (the other lanes hold values nobody reads), sgn = +/-1 as a sign-bit mask.  Lane bits 0..3 go through the DPP
// network, bits 4 / 5 through v_permlane16/32_swap on a copy -- no ds_bpermute round trips through the LDS pipeline.
template <int LB, typename A>
__device__ __forceinline__ A pruned_lane_step(A v, uint32_t sign_bit)
{
    if constexpr (LB < 4) {
        const A t = Bits<A>::fold_sign(v, sign_bit);        // the partner reads +/- v
        return v + Bits<A>::template partner_dpp<LB>(t);
    } else {
        A a = v, b = v;
        swap_pair<(LB == 4) ? 16 : 32>(a, b);               // even rows / lower half: a' = own v, b' = the partner's v
        return a + Bits<A>::fold_sign(b, sign_bit);
    }
}

// +1 or -1 (as A) from bit `bit` of i: the multiplier of the subtracted / added half at that butterfly stage
template <typename A> __device__ __forceinline__ A stage_sign(uint32_t i, int bit)
{
    return ((i >> bit) & 1u) ? (A)-1 : (A)1;
}
// a + s * b with s = +/-1 exactly: one fused instruction, bit-identical to a + b / a - b (s * b is exact)
__device__ __forceinline__ float pm_add(float a, float s, float b) { return __builtin_fmaf(b, s, a); }
__device__ __forceinline__ double pm_add(double a, double s, double b) { return __builtin_fma(b, s, a); }

// Tile ownership and index helpers as in fused_shs_kernel.  Rows are (J, S, R) x D, s1 / s2 are (J, D),
// u is (J, S, D) -- or (J, 1 + S, D) with MEAN, row 0 of each j being the mean vector added to every sample's
// matrix -- and the outputs are laid out like u (first R entries of rows 1.. written; MEAN leaves row 0 to the caller's sum).
//
// Structure (round 2; the round-1 kernel sat at 0.47 of the HBM peak and was VALU-bound: ~1900 VALU instructions per
// 16 KiB tile cap a CU at 1.2 tiles / us = 5.0 TB/s for the chip -- profiles/r02/wbar_bwd_*):
//   * every row's scalars (s1, u, u_mean, s2, the diagonal element of dL/dW, its indices) are fetched ONCE, up front,
//     next to the tile loads -- wave-uniform (scalar loads) for rows of >= 64 chunks;
//   * two butterfly networks, chosen by size (launch_wbar_bwd): POLICY_LDS (f32, 64-register tiles, cache-resident
//     gradients) runs the six lane-bit stages as in-register adds after one transpose through a private LDS slab
//     (fwht_tile_lds: 704 add/sub per tile, 8 waves per CU); POLICY_DPP in its SIGNED form (f32, streams) keeps the
//     DPP / permlane network with one v_fmac_f32_dpp per lane-stage element and 16 waves per CU -- the sign
//     convention it leaves behind costs nothing here, it only flips bits of the row index in the pruned butterfly;
//   * the row sums share ONE signed sum: dL/du_i = sum_d g1[d] (H[i,d] s2_i) and dL/ds2_i = sum_d H[d,i] (u_i g1[d]) are
//     s2_i * c and u_i * c with c = sum_d H[i,d] g1[d] = (H g1)[i], the common factor taken out of the sum (the op chain
//     multiplies every term and then adds: same value up to the rounding of a D-term sum, whose order autograd leaves
//     unspecified anyway).  c is ONE coefficient of a further transform of g1, i.e. a pruned butterfly: at the stage of
//     index bit b keep a + b or a - b according to bit b of i -- D - 1 additions per row instead of 2-3 D multiplies
//     and adds, written as fma(b, +/-1, a) (exact: the same bits as a +/- b) with the sign a wave-uniform scalar;
//   * the lane-bit stages of that pruned butterfly run through DPP / permlane swaps (pruned_lane_step).
// (Tried and dropped: a persistent grid of 2 blocks per CU whose waves prefetch tile t + stride into the registers the
// LDS-limited occupancy leaves free -- 4.1-4.7 vs 4.4-4.9 TB/s at 1 GiB, 5.4 vs 5.6 at 4 GiB: the kernel was bound
// by its VALU work, not by exposed load latency; tools/readbench.hip has the same comparison on a bare read stream.)
template <typename T, int LOG2D, int K, bool NT, bool MEAN, int POLICY>
__global__ void __launch_bounds__(256)
wbar_bwd_kernel(T *grad_u, T *part_s1, T *part_s2, const u32x4 *gw, const T *__restrict__ s1, const T *__restrict__ u,
                const T *__restrict__ s2, int64_t n_chunks, int64_t n_tiles, uint32_t n_rows, FastDiv by_r, FastDiv by_s)
{
    using E = Elem<T>;
    using A = typename E::acc;
    static_assert(sizeof(A) == sizeof(T), "f32 / f64 only");
    constexpr int VEC = E::VEC;
    constexpr int LV = ilog2(VEC);
    constexpr int TILE = 64 * K;
    constexpr int SH = LOG2D - LV;                    // log2(chunks per row)
    constexpr uint32_t CPR = 1u << SH;
    constexpr uint32_t D = 1u << LOG2D;
    static_assert(LOG2D >= LV, "rows of at least one chunk");
    // rows of one tile: SH >= 6 -> every row covers all 64 lanes and KPR = CPR/64 consecutive k;
    //                   SH <  6 -> every k holds 64/CPR rows side by side in the lanes
    constexpr int KPR = SH >= 6 ? (int)(CPR / 64) : 1;
    constexpr int NACC = K / KPR;
    constexpr int LANE_BITS = SH >= 6 ? 6 : SH;
    // DPP network in its signed form for f32 (fwht_tile.hpp): the transform leaves sigma(lane) * g1 with
    // sigma = (-1)^popcount(lane & SIGN_OUT).  No repair is needed here: the pruned butterfly below multiplies lane l's
    // partial sum by (-1)^popcount(i_lane & l) anyway, and sigma just flips the bits of i_lane under SIGN_OUT.
    constexpr bool DPP_SIGNED = POLICY == POLICY_DPP && std::is_same<A, float>::value;
    constexpr int SIGN_OUT = DPP_SIGNED ? fwht_sign_out<VEC, LOG2D>(0) : 0;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t blk = blockIdx.x;
    if (NT && (gridDim.x & 7) == 0) blk = (blk & 7) * (int64_t)(gridDim.x >> 3) + (blk >> 3);   // XCD-contiguous
    const int64_t t = blk * 4 + wave;
    if (t >= n_tiles) return;
    const uint32_t lane_col = (uint32_t)lane & (CPR - 1);        // chunk column of this lane within its row (+ kk * 64)

    auto first_row = [&](int64_t tile, int k) -> uint32_t {
        const uint32_t row0 = (uint32_t)((tile * TILE) >> SH);
        if constexpr (SH >= 6) return row0 + (uint32_t)((k * 64) >> SH);                 // wave-uniform
        else return row0 + (uint32_t)((k * 64 + lane) >> SH);
    };
    // scalar operands of a tile's rows; rows past the end read row 0's (valid memory, results discarded)
    struct Rows {
        uint32_t ri[NACC];           // row index i inside its matrix
        size_t ro[NACC];             // offset of the row's three outputs (laid out like u)
        A uv[NACC], uv0[NACC], s2v[NACC], gii[NACC];
    };
    // rows past the end read row 0's operands (valid memory, results discarded)
    auto row_index = [&](int64_t tile, int n, uint32_t &rr, uint32_t &i, uint32_t &j, uint32_t &ur) {
        const uint32_t row = first_row(tile, n * KPR);
        rr = row < n_rows ? row : 0u;
        const uint32_t jk = by_r.div(rr);
        i = rr - jk * by_r.d;
        j = by_s.div(jk);
        ur = MEAN ? jk + j + 1 : jk;     // MEAN: u is (J, 1 + S, D) = [u_mean; u_1 .. u_S], W[j,k] = w_bar(u_mean) + w_bar(u_k)
    };
    // the tile itself and the one scalar its first step needs
    auto fetch_data = [&](int64_t tile, u32x4 (&raw)[K], A (&s1v)[NACC]) {
        const int64_t base = tile * TILE;
        if (base + TILE <= n_chunks) {               // branch-free loads for full tiles
#pragma unroll
            for (int k = 0; k < K; ++k) raw[k] = ld16<NT>(gw + base + k * 64 + lane);
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                u32x4 z = {0u, 0u, 0u, 0u};
                raw[k] = (base + k * 64 + lane < n_chunks) ? ld16<NT>(gw + base + k * 64 + lane) : z;
            }
        }
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            uint32_t rr, i, j, ur;
            row_index(tile, n, rr, i, j, ur);
            s1v[n] = (A)s1[(size_t)j * D + i];
        }
    };
    // everything the row sums need: requested before the transform, consumed after it
    auto fetch_rows = [&](int64_t tile, Rows &rw) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            uint32_t rr, i, j, ur;
            row_index(tile, n, rr, i, j, ur);
            rw.ri[n] = i;
            rw.ro[n] = (size_t)ur * D + i;
            rw.s2v[n] = (A)s2[(size_t)j * D + i];
            rw.uv[n] = (A)u[(size_t)ur * D + i];
            if constexpr (MEAN) rw.uv0[n] = (A)u[(size_t)(j * by_s.d + j) * D + i];
            else rw.uv0[n] = (A)0;
            rw.gii[n] = (A)reinterpret_cast<const T *>(gw)[(size_t)rr * D + i];   // dL/dW[row, i]: a line this tile loads anyway
        }
    };
    auto scale_in = [&](const u32x4 (&raw)[K], const A (&s1v)[NACC], A (&r)[K][VEC]) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            E::unpack(raw[k], r[k]);
            const A sv = s1v[k / KPR];
#pragma unroll
            for (int e = 0; e < VEC; e += 2) mul2(r[k][e], r[k][e + 1], sv, sv);
        }
    };
    auto transform = [&](A (&r)[K][VEC]) {
        if constexpr (POLICY == POLICY_LDS) {
            extern __shared__ __attribute__((aligned(16))) char whvi_smem[];
            fwht_tile_lds<A, VEC, K, LOG2D>(r, lane, reinterpret_cast<A *>(whvi_smem) + wave * lds_slab_floats<VEC, K>());
        } else if constexpr (DPP_SIGNED) {
            fwht_tile<A, VEC, K, LOG2D, POLICY_DPP, 0, true, 0>(r, lane);       // g1, held as sigma(lane) * g1 (see SIGN_OUT)
        } else {
            fwht_tile<A, VEC, K, LOG2D, POLICY_DPP, 0>(r, lane);       // g1
        }
    };
    // c = (H g1)[i] by a pruned butterfly over the row's elements d = (kk * 64 + lane_col) * VEC + e, then
    // dL/du = s2_i c, dL/ds2 = u_i c (+ u0_i c); one lane per row writes the three results
    auto sums_out = [&](int64_t tile, const Rows &rw, A (&r)[K][VEC]) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            const uint32_t i = rw.ri[n];
            // index bits [0, LV): positions inside a chunk
#pragma unroll
            for (int b = 0; b < LV; ++b) {
                const A sg = stage_sign<A>(i, b);
#pragma unroll
                for (int kk = 0; kk < KPR; ++kk)
#pragma unroll
                    for (int e = 0; e < VEC; e += 2 << b) {
                        A &lo = r[n * KPR + kk][e];
                        lo = pm_add(lo, sg, r[n * KPR + kk][e + (1 << b)]);
                    }
            }
            // index bits [LV + 6, LOG2D): the row's chunks held by this lane (rows of >= 64 chunks only)
#pragma unroll
            for (int b = 0; (1 << b) < KPR; ++b) {
                const A sg = stage_sign<A>(i, LV + 6 + b);
#pragma unroll
                for (int kk = 0; kk < KPR; kk += 2 << b) {
                    A &lo = r[n * KPR + kk][0];
                    lo = pm_add(lo, sg, r[n * KPR + kk + (1 << b)][0]);
                }
            }
            // index bits [LV, LV + LANE_BITS): across the lanes that share the row
            A c = r[n * KPR][0];
            static_for<0, LANE_BITS>([&](auto lb) {
                constexpr int LB = decltype(lb)::value;
                c = pruned_lane_step<LB>(c, (((i >> (LV + LB)) ^ (uint32_t)(SIGN_OUT >> LB)) & 1u) << 31);
            });
            // outputs are laid out like u; entries i >= R stay untouched
            const uint32_t row = first_row(tile, n * KPR);
            const bool writer = (SH >= 6) ? (lane == 0) : (lane_col == 0);
            if (writer && row < n_rows) {
                A p1 = rw.gii[n] * ((A)D * (rw.uv[n] * rw.s2v[n]));
                A p2 = rw.uv[n] * c;
                if constexpr (MEAN) {
                    p1 += rw.gii[n] * ((A)D * (rw.uv0[n] * rw.s2v[n]));
                    p2 += rw.uv0[n] * c;
                }
                grad_u[rw.ro[n]] = (T)(c * rw.s2v[n]);
                part_s2[rw.ro[n]] = (T)p2;
                part_s1[rw.ro[n]] = (T)p1;
            }
        }
    };

    u32x4 raw[K];
    A s1v[NACC];
    Rows rw;
    fetch_data(t, raw, s1v);
    fetch_rows(t, rw);
    A r[K][VEC];
    scale_in(raw, s1v, r);
    transform(r);
    sums_out(t, rw, r);
}

template <typename T, int LOG2D>
inline void launch_wbar_bwd(void *grad_u, void *part_s1, void *part_s2, const void *gw, const void *s1,
                            const void *u, const void *s2, int64_t rows, int64_t S, int64_t R, bool mean, bool no_lds,
                            int small_tiles, hipStream_t st)
{
    constexpr int K = pick_k<T, LOG2D>();
    constexpr int VEC = Elem<T>::VEC;
    using A = typename Elem<T>::acc;
    const int64_t n_chunks = (rows << LOG2D) / VEC;
    // Cache-resident problems (<= the 256 MiB Infinity Cache): 16 KiB tiles give a CU only a handful of waves (config 2's
    // backward, 32 MiB = 2 048 tiles = 8 waves per CU, each doing a whole tile's work back to back).  A quarter-size tile
    // (never less than one row) makes four times the waves with a quarter of the work each.  DPP network (signed for
    // f32), cached loads.  Kernel-trace medians, 16 KiB tiles (LDS-staged <= 128 MiB) -> quarter tiles
    // (profiles/r02/wbar_bwd_tile_size.log): D = 4 x 256 matrices 5.8 -> 3.5 us, D = 512 x 32 (32 MiB) 9.7 -> 8.4,
    // D = 1024 x 16 (64 MiB) 16.4 -> 14.7, D = 512 x 128 (128 MiB) 29.9 -> 22.8, 256 MiB 41-49 -> 41-44; at 512 MiB the
    // two tie and the streaming launch below takes over.
    constexpr int LVc = ilog2(VEC);
    constexpr int NEED = (LOG2D > LVc + 6) ? (1 << (LOG2D - LVc - 6)) : 1;
    constexpr int KS = NEED > 4 ? NEED : 4;
    if constexpr (KS < K) {
        const bool small = small_tiles > 0 || (small_tiles == 0 && n_chunks * 16 <= NT_MIN_BYTES);
        if (small) {
            const int64_t tiles_s = (n_chunks + 64 * KS - 1) / (64 * KS);
            const FastDiv dr_s = make_fastdiv((uint32_t)R), ds_s = make_fastdiv((uint32_t)S);
#define WHVI_BWD_SMALL(MEAN)                                                                                        \
    do {                                                                                                            \
        note_launch<T>("wbar_bwd_kernel", LOG2D, KS, false, (bool)MEAN, (int)POLICY_DPP);                           \
        hipLaunchKernelGGL((wbar_bwd_kernel<T, LOG2D, KS, false, MEAN, POLICY_DPP>), dim3((unsigned)((tiles_s + 3) / 4)), \
                           dim3(256), 0, st, (T *)grad_u, (T *)part_s1, (T *)part_s2, (const u32x4 *)gw, (const T *)s1, \
                           (const T *)u, (const T *)s2, n_chunks, tiles_s, (uint32_t)rows, dr_s, ds_s);             \
    } while (0)
            if (mean) WHVI_BWD_SMALL(true); else WHVI_BWD_SMALL(false);
#undef WHVI_BWD_SMALL
            return;
        }
    }
    const int64_t n_tiles = (n_chunks + 64 * K - 1) / (64 * K);
    const FastDiv dr = make_fastdiv((uint32_t)R), ds = make_fastdiv((uint32_t)S);
    const int64_t blocks = (n_tiles + 3) / 4;
    // the LDS-staged network needs 32-bit arithmetic and a 64-register tile; rows of at least 64 chunks make the
    // transposes worth it (below that few lane-bit stages exist and the DPP network is short)
    constexpr bool LDS_OK = sizeof(A) == 4 && K * VEC == 64 && LOG2D >= 8;
    constexpr size_t slab_bytes = (size_t)K * (64 * VEC + VEC) * 4;
#define WHVI_BWD(NT, MEAN, POL)                                                                                  \
    do {                                                                                                         \
        note_launch<T>("wbar_bwd_kernel", LOG2D, K, (bool)NT, (bool)MEAN, (int)POL);                             \
        hipLaunchKernelGGL((wbar_bwd_kernel<T, LOG2D, K, NT, MEAN, POL>), dim3((unsigned)blocks), dim3(256),     \
                           (POL == POLICY_LDS) ? 4 * slab_bytes : 0, st, (T *)grad_u, (T *)part_s1, (T *)part_s2, \
                           (const u32x4 *)gw, (const T *)s1, (const T *)u, (const T *)s2, n_chunks, n_tiles,     \
                           (uint32_t)rows, dr, ds);                                                              \
    } while (0)
    // Which butterfly network for 16 KiB tiles (measured, profiles/r02/wbar_bwd_lds_vs_dpp.log): gradients of <= 128 MiB
    // that are forced onto big tiles (WHVI_WBAR_BIG_TILES: cross-checks) take the LDS-staged one (fewest instructions
    // per wave: 10.0 vs 11.8 us at 32 MiB, 16.2 vs 17.4 at 64 MiB); streams take the DPP network in its signed form (1 050 VALU instructions per tile instead of 633 + 96 LDS, but 100 VGPRs
    // and no LDS = 16 instead of 8 waves per CU): 6.3-6.4 vs 5.8-6.1 TB/s at 512 MiB, 5.3-6.0 vs 5.1-5.4 at 1 GiB,
    // 6.56-6.62 vs 6.05-6.08 at 4 GiB.
    const bool use_lds = !no_lds && n_chunks * 16 <= ((int64_t)128 << 20);
#define WHVI_BWD_POL(NT, MEAN)                                           \
    do {                                                                 \
        if constexpr (LDS_OK) {                                          \
            if (use_lds) { WHVI_BWD(NT, MEAN, POLICY_LDS); break; }      \
        }                                                                \
        WHVI_BWD(NT, MEAN, POLICY_DPP);                                  \
    } while (0)
    const bool nt = n_chunks * 16 > NT_MIN_BYTES;      // a read-only stream: non-temporal beyond the Infinity Cache
    if (mean) { if (nt) WHVI_BWD_POL(true, true); else WHVI_BWD_POL(false, true); }
    else { if (nt) WHVI_BWD_POL(true, false); else WHVI_BWD_POL(false, false); }
#undef WHVI_BWD_POL
#undef WHVI_BWD
}

template <typename T>
inline int wbar_bwd_dispatch(void *grad_u, void *part_s1, void *part_s2, const void *gw, const void *s1,
                             const void *u, const void *s2, int64_t J, int64_t S, int64_t R, int32_t log2d,
                             int32_t flags, void *stream)
{
    constexpr int LV = ilog2(Elem<T>::VEC);
    g_err[0] = 0;
    if (J < 0 || S < 0 || R < 0) return fail(WHVI_ERR_ARG, "whvi_wbar_bwd: negative size%s", "");
    if (flags & ~(WHVI_WBAR_MEAN | WHVI_WBAR_NO_LDS | WHVI_WBAR_SMALL_TILES | WHVI_WBAR_BIG_TILES)) return fail(WHVI_ERR_ARG, "whvi_wbar_bwd: unknown flags%s 0x%llx", "", flags);
    if (log2d < LV || log2d > max_single_pass_log2d<T>())
        return fail(WHVI_ERR_SIZE, "whvi_wbar_bwd: log2(D)%s = %lld is outside the supported range [%lld, ...]", "",
                    log2d, LV);
    if (R > ((int64_t)1 << log2d)) return fail(WHVI_ERR_ARG, "whvi_wbar_bwd: R%s = %lld exceeds D", "", R);
    const int64_t rows = J * S * R;
    if (rows == 0) return WHVI_OK;
    if (rows >= ((int64_t)1 << 32)) return fail(WHVI_ERR_SIZE, "whvi_wbar_bwd: rows are indexed with 32 bits%s", "");
    if (!grad_u || !part_s1 || !part_s2 || !gw || !s1 || !u || !s2)
        return fail(WHVI_ERR_ARG, "whvi_wbar_bwd: null pointer%s", "");
    if ((uintptr_t)gw & 15) return fail(WHVI_ERR_ALIGN, "whvi_wbar_bwd: %s pointer is not 16-byte aligned", "grad_w");
    hipStream_t st = (hipStream_t)stream;
#define WHVI_CASE(L)                                                                                       \
    case L:                                                                                                \
        if constexpr (L >= LV && L <= max_single_pass_log2d<T>())                                          \
            launch_wbar_bwd<T, L>(grad_u, part_s1, part_s2, gw, s1, u, s2, rows, S, R, (flags & WHVI_WBAR_MEAN) != 0,      \
                                  (flags & WHVI_WBAR_NO_LDS) != 0,                                                       \
                                  (flags & WHVI_WBAR_SMALL_TILES) ? 1 : ((flags & WHVI_WBAR_BIG_TILES) ? -1 : 0), st);                 \
        break;
    switch (log2d) {
        WHVI_CASE(1) WHVI_CASE(2) WHVI_CASE(3) WHVI_CASE(4) WHVI_CASE(5) WHVI_CASE(6) WHVI_CASE(7)
        WHVI_CASE(8) WHVI_CASE(9) WHVI_CASE(10) WHVI_CASE(11) WHVI_CASE(12) WHVI_CASE(13)
    default: break;
    }
#undef WHVI_CASE
    return after_launch("wbar_bwd");
}

}  // namespace whvi
