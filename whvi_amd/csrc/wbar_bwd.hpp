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

template <typename A> __device__ __forceinline__ A flip_if(A v, bool neg) { return neg ? -v : v; }

// Tile ownership and index helpers as in fused_shs_kernel.  Rows are (J, S, R) x D, s1 / s2 are (J, D),
// u is (J, S, D) -- or (J, 1 + S, D) with MEAN, row 0 of each j being the mean vector added to every sample's
// matrix -- and the outputs are laid out like u (first R entries of rows 1.. written; MEAN leaves row 0 to the caller's sum).
template <typename T, int LOG2D, int K, bool NT, bool MEAN>
__global__ void __launch_bounds__(256)
wbar_bwd_kernel(T *grad_u, T *part_s1, T *part_s2, const u32x4 *gw, const T *s1, const T *u, const T *s2,
                int64_t n_chunks, int64_t n_tiles, uint32_t n_rows, FastDiv by_r, FastDiv by_s)
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

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t t = (int64_t)blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;

    const int64_t base = t * TILE;
    const bool full = base + TILE <= n_chunks;
    const uint32_t row0 = (uint32_t)(base >> SH);
    auto chunk_row = [&](int k) -> uint32_t {
        if constexpr (SH >= 6) return row0 + (uint32_t)((k * 64) >> SH);                 // wave-uniform
        else return row0 + (uint32_t)((k * 64 + lane) >> SH);
    };
    auto chunk_col = [&](int k) -> uint32_t { return (uint32_t)(k * 64 + lane) & (CPR - 1); };
    // scalar operands of a row; rows past the end read row 0's (valid memory, results discarded)
    struct RowIdx { uint32_t i, jk, j, ur, u0; };   // row i of matrix (j, k); jk = j * S + k; ur / u0: rows of u
    auto row_index = [&](uint32_t row) -> RowIdx {
        const uint32_t rr = row < n_rows ? row : 0u;
        const uint32_t jk = by_r.div(rr);
        const uint32_t i = rr - jk * by_r.d;
        const uint32_t j = by_s.div(jk);
        // MEAN: u is (J, 1 + S, D) = [u_mean; u_1 .. u_S] and W[j,k] = w_bar(u_mean) + w_bar(u_k)
        return RowIdx{i, jk, j, MEAN ? jk + j + 1 : jk, MEAN ? j * by_s.d + j : 0u};
    };

    A r[K][VEC];
    {
        u32x4 raw[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            u32x4 z = {0u, 0u, 0u, 0u};
            raw[k] = (full || base + k * 64 + lane < n_chunks) ? ld16<NT>(gw + base + k * 64 + lane) : z;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const RowIdx x = row_index(chunk_row(k));
            const A s1v = (A)s1[(size_t)x.j * D + x.i];
            E::unpack(raw[k], r[k]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) r[k][e] = s1v * r[k][e];
        }
    }
    fwht_tile<A, VEC, K, LOG2D, POLICY_DPP, 0>(r, lane);       // g1

    A acc_u[NACC], acc_s2[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
        const RowIdx x = row_index(chunk_row(n * KPR));
        const A uv = (A)u[(size_t)x.ur * D + x.i];
        const A uv0 = MEAN ? (A)u[(size_t)x.u0 * D + x.i] : (A)0;
        const A s2v = (A)s2[(size_t)x.j * D + x.i];
        A su = (A)0, ss = (A)0;
        // H[i,d] = (-1)^popcount(i & d) with d = dbase + e, dbase a multiple of VEC: the parity splits into one term
        // per chunk and one per in-chunk position, and the sign goes onto g1 once ((-g) * s == -(g * s) exactly)
        bool par_e[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) par_e[e] = __builtin_popcount(x.i & (uint32_t)e) & 1;
#pragma unroll
        for (int kk = 0; kk < KPR; ++kk) {
            const int k = n * KPR + kk;
            const bool par_k = __builtin_popcount(x.i & (chunk_col(k) * VEC)) & 1;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const A g = flip_if(r[k][e], par_k != par_e[e]);
                su += g * s2v;
                ss += uv * g;
                if constexpr (MEAN) ss += uv0 * g;
            }
        }
#pragma unroll
        for (int lb = 0; lb < LANE_BITS; ++lb) {
            su += __shfl_xor(su, 1 << lb, 64);
            ss += __shfl_xor(ss, 1 << lb, 64);
        }
        acc_u[n] = su;
        acc_s2[n] = ss;
    }
    // one lane per row writes the three results
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
        const uint32_t row = chunk_row(n * KPR);
        const bool writer = (SH >= 6) ? (lane == 0) : ((lane & (int)(CPR - 1)) == 0);
        if (writer && row < n_rows) {
            const RowIdx x = row_index(row);
            const A uv = (A)u[(size_t)x.ur * D + x.i];
            const A s2v = (A)s2[(size_t)x.j * D + x.i];
            const A gii = (A)reinterpret_cast<const T *>(gw)[(size_t)row * D + x.i];
            const size_t o = (size_t)x.ur * D + x.i;          // outputs are laid out like u; entries i >= R stay untouched
            A p1 = gii * ((A)D * (uv * s2v));
            if constexpr (MEAN) p1 += gii * ((A)D * ((A)u[(size_t)x.u0 * D + x.i] * s2v));
            grad_u[o] = (T)acc_u[n];
            part_s2[o] = (T)acc_s2[n];
            part_s1[o] = (T)p1;
        }
    }
}

template <typename T, int LOG2D>
inline void launch_wbar_bwd(void *grad_u, void *part_s1, void *part_s2, const void *gw, const void *s1,
                            const void *u, const void *s2, int64_t rows, int64_t S, int64_t R, bool mean, hipStream_t st)
{
    constexpr int K = pick_k<T, LOG2D>();
    constexpr int VEC = Elem<T>::VEC;
    const int64_t n_chunks = (rows << LOG2D) / VEC;
    const int64_t n_tiles = (n_chunks + 64 * K - 1) / (64 * K);
    const FastDiv dr = make_fastdiv((uint32_t)R), ds = make_fastdiv((uint32_t)S);
    const unsigned grid = (unsigned)((n_tiles + 3) / 4);
#define WHVI_BWD(NT, MEAN)                                                                                       \
    hipLaunchKernelGGL((wbar_bwd_kernel<T, LOG2D, K, NT, MEAN>), dim3(grid), dim3(256), 0, st, (T *)grad_u,      \
                       (T *)part_s1, (T *)part_s2, (const u32x4 *)gw, (const T *)s1, (const T *)u,         \
                       (const T *)s2, n_chunks, n_tiles, (uint32_t)rows, dr, ds)
    const bool nt = n_chunks * 16 >= NT_MIN_BYTES;
    if (mean) { if (nt) WHVI_BWD(true, true); else WHVI_BWD(false, true); }
    else { if (nt) WHVI_BWD(true, false); else WHVI_BWD(false, false); }
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
    if (flags & ~WHVI_WBAR_MEAN) return fail(WHVI_ERR_ARG, "whvi_wbar_bwd: unknown flags%s 0x%llx", "", flags);
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
            launch_wbar_bwd<T, L>(grad_u, part_s1, part_s2, gw, s1, u, s2, rows, S, R, (flags & WHVI_WBAR_MEAN) != 0, st);                 \
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
