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
// Every product is its own rounding, as in the separate ATen kernels; the two row sums run as an in-lane
// sum followed by an xor-shuffle tree (autograd's own reduction order is unspecified as well).
#include "kernels.hpp"

namespace whvi {

// One xor-butterfly step of a wave all-reduce: every lane ends up with v(lane) + v(lane ^ (1 << LB)).  Lane bits 0..3
// go through the DPP network, bits 4 / 5 through v_permlane16/32_swap on a copy (a' + b' holds the sum of the two
// 16-lane rows / 32-lane halves in every lane) -- no ds_bpermute round trips through the LDS pipeline.
template <int LB, typename A>
__device__ __forceinline__ A xor_reduce_step(A v)
{
    if constexpr (LB < 4) {
        return v + Bits<A>::template partner_dpp<LB>(v);
    } else {
        A a = v, b = v;
        swap_pair<(LB == 4) ? 16 : 32>(a, b);
        return a + b;
    }
}

template <typename A> __device__ __forceinline__ A sign_flip(A v, uint32_t sign_bit)    // sign_bit: 0 or 0x80000000
{
    return Bits<A>::fold_sign(v, sign_bit);
}
__device__ __forceinline__ uint32_t parity_sign(uint32_t x) { return (uint32_t)(__builtin_popcount(x) & 1) << 31; }

// Tile ownership and index helpers as in fused_shs_kernel.  Rows are (J, S, R) x D, s1 / s2 are (J, D),
// u is (J, S, D) -- or (J, 1 + S, D) with MEAN, row 0 of each j being the mean vector added to every sample's
// matrix -- and the outputs are laid out like u (first R entries of rows 1.. written; MEAN leaves row 0 to the caller's sum).
//
// Structure (round 2; the round-1 kernel sat at 0.47 of the HBM peak and was VALU-bound: ~1900 VALU instructions per
// 16 KiB tile cap a CU at 1.2 tiles / us = 5.0 TB/s for the chip -- profiles/r02/wbar_bwd_*):
//   * every row's scalars (s1, u, u_mean, s2, the diagonal element of dL/dW, its indices) are fetched ONCE, up front,
//     next to the tile loads -- wave-uniform (scalar loads) for rows of >= 64 chunks;
//   * POLICY_LDS (f32, 64-register tiles): the six lane-bit stages of the transform run as packed in-register adds
//     after one transpose through a private LDS slab (fwht_tile_lds: a third of the DPP network's issue slots);
//   * the three signed row sums share one pass: the Hadamard sign (-1)^popcount(i & d) splits into an in-chunk part
//     folded into the multipliers (+/- s2_i, +/- u_i, +/- u0_i per position: (-s) * g == -(s * g) exactly), a per-chunk
//     part applied to the chunk's partial sums and a per-lane part applied once before the cross-lane reduction;
//     products stay separate roundings (built with -ffp-contract=off), packed two per instruction;
//   * cross-lane sums through DPP / permlane swaps (xor_reduce_step).
//   * PIPE (LDS policy, big problems): the LDS slabs cap a CU at 8 waves, far fewer than its registers allow, so the
//     spare registers hold the NEXT tile: a persistent grid (2 blocks per CU) walks the tiles with a grid stride and
//     every wave issues tile t + stride's loads (data and row scalars) before it transforms tile t.
template <typename T, int LOG2D, int K, bool NT, bool MEAN, int POLICY, bool PIPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PIPE ? 2 : 1)))
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
    static_assert(!PIPE || SH >= 6, "the pipelined form keeps the next tile's row scalars in SGPRs: wave-uniform rows");
    // rows of one tile: SH >= 6 -> every row covers all 64 lanes and KPR = CPR/64 consecutive k;
    //                   SH <  6 -> every k holds 64/CPR rows side by side in the lanes
    constexpr int KPR = SH >= 6 ? (int)(CPR / 64) : 1;
    constexpr int NACC = K / KPR;
    constexpr int LANE_BITS = SH >= 6 ? 6 : SH;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t blk = blockIdx.x;
    if (NT && !PIPE && (gridDim.x & 7) == 0) blk = (blk & 7) * (int64_t)(gridDim.x >> 3) + (blk >> 3);   // XCD-contiguous
    int64_t t = blk * 4 + wave;
    if (t >= n_tiles) return;
    const int64_t stride = (int64_t)gridDim.x * 4;
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
        const bool full = base + TILE <= n_chunks;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            u32x4 z = {0u, 0u, 0u, 0u};
            raw[k] = (full || base + k * 64 + lane < n_chunks) ? ld16<NT>(gw + base + k * 64 + lane) : z;
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
        } else {
            fwht_tile<A, VEC, K, LOG2D, POLICY_DPP, 0>(r, lane);       // g1
        }
    };
    // the row sums  dL/du = sum_d g1[d] * (H[i,d] s2_i)  and  dL/ds2 = sum_d H[d,i] * (u_i g1[d]) (+ u0_i g1[d]),
    // H[i,d] = (-1)^popcount(i & d) with d = (kk * 64 + lane_col) * VEC + e; then one lane per row writes the results
    auto sums_out = [&](int64_t tile, const Rows &rw, A (&r)[K][VEC]) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            const uint32_t i = rw.ri[n];
            A m1[VEC], m2[VEC], m3[VEC];          // multipliers with the in-chunk sign folded in
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const uint32_t se = parity_sign(i & (uint32_t)e);
                m1[e] = sign_flip(rw.s2v[n], se);
                m2[e] = sign_flip(rw.uv[n], se);
                m3[e] = sign_flip(rw.uv0[n], se);
            }
            A su[2] = {(A)0, (A)0}, ss[2] = {(A)0, (A)0};
#pragma unroll
            for (int kk = 0; kk < KPR; ++kk) {
                const int k = n * KPR + kk;
                A pu[2] = {(A)0, (A)0}, ps[2] = {(A)0, (A)0};
#pragma unroll
                for (int e = 0; e < VEC; e += 2) {
                    A a0 = r[k][e], a1 = r[k][e + 1];
                    mul2(a0, a1, m1[e], m1[e + 1]);                      // g * (+/- s2_i)
                    add2(pu[0], pu[1], a0, a1);
                    A b0 = r[k][e], b1 = r[k][e + 1];
                    mul2(b0, b1, m2[e], m2[e + 1]);                      // (+/- u_i) * g
                    add2(ps[0], ps[1], b0, b1);
                    if constexpr (MEAN) {
                        A c0 = r[k][e], c1 = r[k][e + 1];
                        mul2(c0, c1, m3[e], m3[e + 1]);                  // (+/- u0_i) * g
                        add2(ps[0], ps[1], c0, c1);
                    }
                }
                if constexpr (KPR > 1) {
                    const uint32_t sk = parity_sign(i & ((uint32_t)kk * 64u * VEC));       // wave-uniform
                    add2(su[0], su[1], sign_flip(pu[0], sk), sign_flip(pu[1], sk));
                    add2(ss[0], ss[1], sign_flip(ps[0], sk), sign_flip(ps[1], sk));
                } else {
                    su[0] = pu[0]; su[1] = pu[1]; ss[0] = ps[0]; ss[1] = ps[1];
                }
            }
            const uint32_t sl = parity_sign(i & (lane_col * VEC));
            A tu = sign_flip(su[0] + su[1], sl), ts = sign_flip(ss[0] + ss[1], sl);
            static_for<0, LANE_BITS>([&](auto lb) {
                tu = xor_reduce_step<decltype(lb)::value>(tu);
                ts = xor_reduce_step<decltype(lb)::value>(ts);
            });
            // outputs are laid out like u; entries i >= R stay untouched
            const uint32_t row = first_row(tile, n * KPR);
            const bool writer = (SH >= 6) ? (lane == 0) : (lane_col == 0);
            if (writer && row < n_rows) {
                A p1 = rw.gii[n] * ((A)D * (rw.uv[n] * rw.s2v[n]));
                if constexpr (MEAN) p1 += rw.gii[n] * ((A)D * (rw.uv0[n] * rw.s2v[n]));
                grad_u[rw.ro[n]] = (T)tu;
                part_s2[rw.ro[n]] = (T)ts;
                part_s1[rw.ro[n]] = (T)p1;
            }
        }
    };

    u32x4 raw[K];
    A s1v[NACC];
    Rows rw;
    fetch_data(t, raw, s1v);
    if constexpr (!PIPE) {
        fetch_rows(t, rw);
        A r[K][VEC];
        scale_in(raw, s1v, r);
        transform(r);
        sums_out(t, rw, r);
    } else {
        for (;;) {
            A r[K][VEC];
            scale_in(raw, s1v, r);
            fetch_rows(t, rw);
            const int64_t tn = t + stride;
            if (tn < n_tiles) fetch_data(tn, raw, s1v);      // in flight while tile t is transformed and summed
            transform(r);
            sums_out(t, rw, r);
            if (tn >= n_tiles) break;
            t = tn;
        }
    }
}

template <typename T, int LOG2D>
inline void launch_wbar_bwd(void *grad_u, void *part_s1, void *part_s2, const void *gw, const void *s1,
                            const void *u, const void *s2, int64_t rows, int64_t S, int64_t R, bool mean, bool no_lds,
                            bool no_pipe, hipStream_t st)
{
    constexpr int K = pick_k<T, LOG2D>();
    constexpr int VEC = Elem<T>::VEC;
    using A = typename Elem<T>::acc;
    const int64_t n_chunks = (rows << LOG2D) / VEC;
    const int64_t n_tiles = (n_chunks + 64 * K - 1) / (64 * K);
    const FastDiv dr = make_fastdiv((uint32_t)R), ds = make_fastdiv((uint32_t)S);
    const int64_t blocks = (n_tiles + 3) / 4;
    // the LDS-staged network needs 32-bit arithmetic and a 64-register tile; rows of at least 64 chunks make the
    // transposes worth it (below that few lane-bit stages exist and the DPP network is short)
    constexpr bool LDS_OK = sizeof(A) == 4 && K * VEC == 64 && LOG2D >= 8;
    constexpr size_t slab_bytes = (size_t)K * (64 * VEC + VEC) * 4;
    // rows of >= 2048 elements: at most two rows per tile, so the next tile's scalars fit the SGPR file next to the
    // current one's (D = 512 / 1024 would spill: tools/check_spills.py)
    constexpr bool PIPE_OK = LDS_OK && LOG2D >= 11;
#define WHVI_BWD(NT, MEAN, POL, PIPE, GRID)                                                                      \
    do {                                                                                                         \
        note_launch<T>("wbar_bwd_kernel", LOG2D, K, (bool)NT, (bool)MEAN, (int)POL, (bool)PIPE);                 \
        hipLaunchKernelGGL((wbar_bwd_kernel<T, LOG2D, K, NT, MEAN, POL, PIPE>), dim3((unsigned)(GRID)), dim3(256), \
                           (POL == POLICY_LDS) ? 4 * slab_bytes : 0, st, (T *)grad_u, (T *)part_s1, (T *)part_s2, \
                           (const u32x4 *)gw, (const T *)s1, (const T *)u, (const T *)s2, n_chunks, n_tiles,     \
                           (uint32_t)rows, dr, ds);                                                              \
    } while (0)
    // pipelined persistent grid: 2 blocks per CU (what the 4 x 16.6 KB slabs of a block allow), once every wave has
    // at least 2 tiles to walk
    const int64_t persistent = (int64_t)num_cu() * 2;
#define WHVI_BWD_POL(NT, MEAN)                                                             \
    do {                                                                                   \
        if constexpr (LDS_OK) {                                                            \
            if (!no_lds) {                                                                 \
                if constexpr (PIPE_OK) {                                                   \
                    if (!no_pipe && blocks >= 2 * persistent) {                            \
                        WHVI_BWD(NT, MEAN, POLICY_LDS, true, persistent);                  \
                        break;                                                             \
                    }                                                                      \
                }                                                                          \
                WHVI_BWD(NT, MEAN, POLICY_LDS, false, blocks);                             \
                break;                                                                     \
            }                                                                              \
        }                                                                                  \
        WHVI_BWD(NT, MEAN, POLICY_DPP, false, blocks);                                     \
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
    if (flags & ~(WHVI_WBAR_MEAN | WHVI_WBAR_NO_LDS | WHVI_WBAR_NO_PIPE)) return fail(WHVI_ERR_ARG, "whvi_wbar_bwd: unknown flags%s 0x%llx", "", flags);
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
                                  (flags & WHVI_WBAR_NO_LDS) != 0, (flags & WHVI_WBAR_NO_PIPE) != 0, st);                 \
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
