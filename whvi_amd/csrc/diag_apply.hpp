#pragma once
// whvi_amd/csrc/diag_apply.hpp -- `h @ (w_bar(g_mu) + w_bar(g_sigma * eps_k)).T` of src/weights.py:87-93 WITHOUT the matrix.
//
// As written in the reference, w_bar(u) = S1 . fwht(diag(u) . fwht(diag(s2))) scales ROWS around row transforms, so it is
// EXACTLY D * diag(s1 (.) u (.) s2) (SURVEY.md finding 1): row i of diag(s2) is one-hot, its transform is s2_i * H[i,:]
// (exact), the row scale makes it +/- v with v = u_i * s2_i (one rounding), the second transform adds those up to D * v at
// column i (exact doublings) and to exact zeros elsewhere, and s1_i scales the row (one rounding).  The dense product with that
// matrix therefore adds exact zeros to ONE product per output: for finite operands `h @ W.T` == h (.) diag(W) + 0, bit for
// bit (the + 0 is the accumulator's: it turns a product of -0 into +0).  These kernels compute the diagonal with the same roundings in the same order
// (wbar_diag below) and apply it as one read + one write of the activations -- no D x D matrix per sample, no GEMM -- and
// they reproduce what the matrix route does with NON-FINITE operands, which is where "multiply by the diagonal" and
// "multiply by a matrix with zeros" differ:
//   * a row of h with a non-finite entry at column j: every OTHER output of that row is NaN (inf * 0, NaN * 0 in the dot
//     products); output j itself is the plain product;
//   * a non-finite s1_i, or a v whose partial sums 2^k v overflow before the last butterfly stage (inf - inf among the
//     off-diagonals, s1_i * 0 with s1_i = inf): row i of W holds NaNs, so output column i is NaN for every row of h.
// Backward: closed form of the same expression (grad_h = g (.) w, grad_w[k,i] = sum_b g (.) h), deterministic summation
// order (rows of a slab in order per thread, row groups through LDS in order, slabs in order in the finishing kernel).
#include "dispatch.hpp"

#ifndef WHVI_DIAG_BWD_UNR
#define WHVI_DIAG_BWD_UNR 4
#endif

namespace whvi {

// diag(w_bar(u))[i], the factor `h @ w_bar(u).T` multiplies h[:, i] by (see the header comment for the NaN rule)
template <typename A, int LOG2D>
__device__ __forceinline__ A wbar_diag(A s1, A s2, A u)
{
    const A v = u * s2;                                   // matmul_diag_left(u, fwht(diag(s2))): +/- (u_i * s2_i)
    A p = s1 * ((A)(1u << LOG2D) * v);                    // the second transform's D * v (exact or inf), then the s1 row scale
    if constexpr (LOG2D >= 1) {
        const A half = (A)(1u << (LOG2D - 1)) * v;        // the largest partial sum below the last stage
        if (!__builtin_isfinite(half) || !__builtin_isfinite(s1)) p = __builtin_nan("");
    }
    return p;
}

// torch.relu / its backward as the fused neighbours compute them (clamp_min(0): NaN stays NaN; threshold_backward: the gradient
// passes unless the activation's RESULT is <= 0, so it also passes where that result is NaN)
template <typename A> __device__ __forceinline__ A relu_(A v) { return (v > (A)0 || v != v) ? v : (A)0; }
constexpr uint32_t DIAG_OPT_MEAN = 1u, DIAG_OPT_RELU_IN = 2u, DIAG_OPT_RELU_OUT = 4u, DIAG_OPT_PLAIN_ORDER = 8u, DIAG_OPT_PLAIN_STORES = 16u;

// streaming store next to streaming loads: write-through + non-temporal (kernels.hpp: the written lines must not displace the
// reads in L2), as a global store (per-lane addresses)
__device__ __forceinline__ void st16_stream(u32x4 *p, const u32x4 &v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <typename T> struct Chunk { typedef typename Elem<T>::acc type __attribute__((ext_vector_type(Elem<T>::VEC))); };

// One 16-byte chunk (VEC columns starting at col) of w_k = w_bar(u_0) + w_bar(u_{first + k}) -- or w_bar(u_k) alone.
template <typename T, int LOG2D>
__device__ __forceinline__ void diag_w_chunk(const T *s1, const T *s2, const T *u, uint32_t sample, uint32_t mean_plus,
                                             uint32_t col, typename Elem<T>::acc (&out)[Elem<T>::VEC])
{
    using A = typename Elem<T>::acc;
    constexpr int VEC = Elem<T>::VEC;
    typedef typename Chunk<T>::type chunk_t;
    const chunk_t a = *reinterpret_cast<const chunk_t *>(s1 + col);
    const chunk_t c = *reinterpret_cast<const chunk_t *>(s2 + col);
    const chunk_t uk = *reinterpret_cast<const chunk_t *>(u + ((size_t)(sample + mean_plus) << LOG2D) + col);
    if (mean_plus) {
        const chunk_t u0 = *reinterpret_cast<const chunk_t *>(u + col);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            out[e] = wbar_diag<A, LOG2D>(a[e], c[e], u0[e]) + wbar_diag<A, LOG2D>(a[e], c[e], uk[e]);   // src/weights.py:93
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) out[e] = wbar_diag<A, LOG2D>(a[e], c[e], uk[e]);
    }
}

// out[k, b, :] = x[(k,) b, :] (.) w_k + bias.  Rows are (S, B) x D; tile ownership as in fwht_rows_kernel (one wave owns
// 64 * K consecutive chunks = whole rows, so the non-finite census of a row is a wave-level reduction).  A block whose rows
// belong to ONE sample computes that sample's w (and the bias) once into LDS; blocks that straddle samples compute w per chunk.
template <typename T, int LOG2D, int K, bool NT, bool XSHARED>
__global__ void __launch_bounds__(256)
diag_apply_kernel(u32x4 *__restrict__ dst, const u32x4 *x, const T *__restrict__ s1, const T *__restrict__ s2,
                  const T *__restrict__ u, const T *__restrict__ bias, int64_t n_chunks, int64_t n_tiles, uint32_t n_rows,
                  FastDiv by_batch, uint32_t opts, uint32_t sample_fastest)
{
    using E = Elem<T>;
    using A = typename E::acc;
    static_assert(sizeof(A) == sizeof(T), "f32 / f64 only");
    constexpr int VEC = E::VEC;
    constexpr int LV = ilog2(VEC);
    constexpr int TILE = 64 * K;
    const uint32_t mean_plus = opts & DIAG_OPT_MEAN;
    const bool relu_in = (opts & DIAG_OPT_RELU_IN) != 0, relu_out = (opts & DIAG_OPT_RELU_OUT) != 0;
    constexpr int SH = LOG2D - LV;
    constexpr uint32_t CPR = 1u << SH;
    constexpr uint32_t D = 1u << LOG2D;
    static_assert(LOG2D >= LV, "rows of at least one chunk");
    static_assert(2 * D * sizeof(A) <= 48 * 1024, "w and the bias of one sample in LDS");
    typedef typename Chunk<T>::type chunk_t;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t blk = blockIdx.x;
    if (XSHARED && sample_fastest != 0u) {
        // shared input (host: whole blocks per sample, blocks per sample a multiple of 8 = sample_fastest * 8): blocks are dealt
        // round-robin over the 8 XCDs; XCD c takes row groups c, c + 8, ... of the input and runs each of them for ALL samples
        // back to back, so an input tile is fetched into that XCD's L2 once and hit S - 1 times.  (Sample-major order re-reads
        // the whole input per sample through the fabric -- as many bytes in as out: 3.4 TB/s written at D = 1024, 16 x 8192 rows)
        const uint32_t b = blockIdx.x, xcd = b & 7u, i = b >> 3;
        const uint32_t n_samples = n_rows / by_batch.d;
        const uint32_t q = i / n_samples, smp = i - q * n_samples;
        blk = (int64_t)smp * (sample_fastest * 8u) + (q * 8u + xcd);
    } else
    if (NT && (gridDim.x & 7) == 0) blk = (blk & 7) * (int64_t)(gridDim.x >> 3) + (blk >> 3);   // XCD-contiguous
    const int64_t t = blk * 4 + wave;
    const bool active = t < n_tiles;
    const int64_t tile0 = t * TILE;
    const bool full = tile0 + TILE <= n_chunks;
    const uint32_t row0 = (uint32_t)(tile0 >> SH);
    auto chunk_row = [&](int k) __attribute__((always_inline)) -> uint32_t {
        if constexpr (SH >= 6) return row0 + (uint32_t)((k * 64) >> SH);
        else return row0 + (uint32_t)((k * 64 + lane) >> SH);
    };
    auto chunk_col = [&](int k) __attribute__((always_inline)) -> uint32_t { return (uint32_t)(k * 64 + lane) & (CPR - 1); };

    // the block's first and last row (its first tile always exists)
    const uint32_t brow0 = (uint32_t)((blk * 4 * TILE) >> SH);
    uint32_t brow1 = (uint32_t)(((blk * 4 + 4) * TILE - 1) >> SH);
    if (brow1 >= n_rows) brow1 = n_rows - 1;
    const uint32_t smp0 = by_batch.div(brow0);
    const bool one_sample = by_batch.div(brow1) == smp0;                 // block-uniform

    // ---- the tile first (HBM / fabric latency), the sample's vectors behind it
    u32x4 raw[K];
    if (active) {
        if constexpr (XSHARED) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t row = chunk_row(k);
                raw[k] = u32x4{0u, 0u, 0u, 0u};
                if (row < n_rows) raw[k] = ld16<false>(x + (int64_t)by_batch.mod(row) * CPR + chunk_col(k));   // cached: re-read by every sample
            }
        } else if (full) {
#pragma unroll
            for (int k = 0; k < K; ++k) raw[k] = ld16<NT>(x + tile0 + k * 64 + lane);
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                raw[k] = u32x4{0u, 0u, 0u, 0u};
                if (tile0 + k * 64 + lane < n_chunks) raw[k] = ld16<false>(x + tile0 + k * 64 + lane);
            }
        }
    }
    __shared__ __attribute__((aligned(16))) A lds_w[D];
    __shared__ __attribute__((aligned(16))) A lds_b[D];
    if (one_sample) {
        constexpr int ITER = (CPR + 255) / 256;
#pragma unroll
        for (int j = 0; j < ITER; ++j) {
            const uint32_t c = threadIdx.x + j * 256;
            if (c < CPR) {
                A wv[VEC];
                diag_w_chunk<T, LOG2D>(s1, s2, u, smp0, mean_plus, c * VEC, wv);
                chunk_t wc, bc;
#pragma unroll
                for (int e = 0; e < VEC; ++e) { wc[e] = wv[e]; bc[e] = (A)0; }
                if (bias != nullptr) bc = *reinterpret_cast<const chunk_t *>(bias + c * VEC);
                *reinterpret_cast<chunk_t *>(lds_w + c * VEC) = wc;
                *reinterpret_cast<chunk_t *>(lds_b + c * VEC) = bc;
            }
        }
        __syncthreads();
    }
    if (!active) {
        if constexpr (NT) __syncthreads();      // the store-alignment barrier below
        return;
    }

    A r[K][VEC];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        E::unpack(raw[k], r[k]);
        if (relu_in) {                          // the activation in FRONT of this layer, applied on load (nn.ReLU fused in)
#pragma unroll
            for (int e = 0; e < VEC; ++e) r[k][e] = relu_(r[k][e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) bad |= !__builtin_isfinite(r[k][e]);
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0) {
        // (rare) some row of the tile holds a non-finite value: every OTHER element of that row becomes NaN -- in the matrix
        // product it meets an exact zero of W (inf * 0, NaN * 0).  Poisoning the input is enough: NaN * w + bias = NaN.
        uint32_t cnt[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            cnt[k] = 0;
#pragma unroll
            for (int e = 0; e < VEC; ++e) cnt[k] += __builtin_isfinite(r[k][e]) ? 0u : 1u;
        }
        if constexpr (SH >= 6) {
            constexpr int KPR = (int)CPR / 64;                    // k-steps per row; K is a multiple of it (pick_k)
#pragma unroll
            for (int j = 0; j < K / KPR; ++j) {
                uint32_t s = 0;
#pragma unroll
                for (int k = j * KPR; k < (j + 1) * KPR; ++k) s += cnt[k];
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) s += (uint32_t)__shfl_xor((int)s, m, 64);
#pragma unroll
                for (int k = j * KPR; k < (j + 1) * KPR; ++k)
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (s - (__builtin_isfinite(r[k][e]) ? 0u : 1u) != 0u) r[k][e] = __builtin_nan("");
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                uint32_t s = cnt[k];
#pragma unroll
                for (int m = 1; m < (int)CPR; m <<= 1) s += (uint32_t)__shfl_xor((int)s, m, 64);   // the CPR lanes of the row
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (s - (__builtin_isfinite(r[k][e]) ? 0u : 1u) != 0u) r[k][e] = __builtin_nan("");
            }
        }
    }

#pragma unroll
    for (int k = 0; k < K; ++k) {
        A wv[VEC], bv[VEC];
        if (one_sample) {
            const chunk_t wc = *reinterpret_cast<const chunk_t *>(lds_w + chunk_col(k) * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { wv[e] = wc[e]; bv[e] = (A)0; }
            if (bias != nullptr) {                                 // (no bias: no second LDS read per chunk)
                const chunk_t bc = *reinterpret_cast<const chunk_t *>(lds_b + chunk_col(k) * VEC);
#pragma unroll
                for (int e = 0; e < VEC; ++e) bv[e] = bc[e];
            }
        } else {
            const uint32_t row = chunk_row(k) < n_rows ? chunk_row(k) : n_rows - 1;      // rows past the end: valid operands, never stored
            diag_w_chunk<T, LOG2D>(s1, s2, u, by_batch.div(row), mean_plus, chunk_col(k) * VEC, wv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) bv[e] = (A)0;
            if (bias != nullptr) {
                const chunk_t bc = *reinterpret_cast<const chunk_t *>(bias + chunk_col(k) * VEC);
#pragma unroll
                for (int e = 0; e < VEC; ++e) bv[e] = bc[e];
            }
        }
        // the one non-zero product of the dot product, added to the +0 the other D - 1 products (exact zeros) sum to in a
        // GEMM's +0-initialised accumulator: a product of -0 comes out as +0 like there (x + 0.0 is not foldable under IEEE rules)
#pragma unroll
        for (int e = 0; e < VEC; ++e) r[k][e] = r[k][e] * wv[e] + (A)0;
        if (bias != nullptr) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) r[k][e] = r[k][e] + bv[e];      // `out + self.bias`, src/weights.py:101-102
        }
        if (relu_out) {                         // the activation BEHIND this layer, applied before the store
#pragma unroll
            for (int e = 0; e < VEC; ++e) r[k][e] = relu_(r[k][e]);
        }
    }
    if constexpr (NT) __syncthreads();          // the block's 4 waves write their 64 KiB back together
    if (NT && full) {
        if constexpr (XSHARED) {                // write-only stream: back-to-back non-temporal global stores (wbar_fwd.hpp)
#pragma unroll
            for (int k = 0; k < K; ++k) st16<true>(dst + tile0 + k * 64 + lane, E::pack(r[k]));
        } else {                                // read + write stream: write-through non-temporal buffer stores, one issue
#pragma unroll                                  // slot apart (kernels.hpp: back to back they cost the headline stream 9 %)
            for (int k = 0; k < K; ++k) {
                tile_store_stream(dst + tile0, lane, k, E::pack(r[k]), TILE * 16);
                asm volatile("s_nop 0");
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (full || tile0 + k * 64 + lane < n_chunks) st16<false>(dst + tile0 + k * 64 + lane, E::pack(r[k]));
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------------
// grid (n_slabs, S): block (slab, k) walks rows [slab * slab_rows, ...) of sample k.  Thread layout: a thread owns CPT
// column chunks (TPR threads cover a row, RG = 256 / TPR rows are in flight per step), so w is computed once per thread and
// the batch reduction grad_w[k, :] = sum_b g (.) x runs in registers.  part: (S, n_slabs, 2, D) = {sum g x, sum g}.
template <typename T, int LOG2D> struct DiagBwdGeom {
    static constexpr int VEC = Elem<T>::VEC;
    static constexpr int CPR = 1 << (LOG2D - ilog2(VEC));
    static constexpr int TPR = CPR < 256 ? CPR : 256;
    static constexpr int CPT = CPR / TPR;
    static constexpr int RG = 256 / TPR;
    static constexpr int UNR = CPT >= 4 ? 2 : (CPT == 1 ? WHVI_DIAG_BWD_UNR : 4);      // rows in flight per thread
};

template <typename T, int LOG2D, bool NT, bool XSHARED, bool WANT_GX>
__global__ void __launch_bounds__(256)
diag_apply_bwd_kernel(u32x4 *__restrict__ gx, T *__restrict__ part, const u32x4 *__restrict__ g, const u32x4 *__restrict__ x,
                      const T *__restrict__ s1, const T *__restrict__ s2, const T *__restrict__ u, const T *__restrict__ bias,
                      uint32_t B, uint32_t slab_rows, uint32_t n_slabs, uint32_t opts)
{
    const uint32_t mean_plus = opts & DIAG_OPT_MEAN;
    const bool relu_in = (opts & DIAG_OPT_RELU_IN) != 0, relu_out = (opts & DIAG_OPT_RELU_OUT) != 0;
    using E = Elem<T>;
    using A = typename E::acc;
    using G = DiagBwdGeom<T, LOG2D>;
    constexpr int VEC = G::VEC, CPR = G::CPR, TPR = G::TPR, CPT = G::CPT, RG = G::RG, UNR = G::UNR;
    constexpr uint32_t D = 1u << LOG2D;
    // 1-D grid of S * n_slabs blocks; streams: XCD-contiguous order (blocks that share an XCD walk one contiguous eighth of
    // the (S, B, D) buffers), a bijection whenever the grid is a multiple of 8
    uint32_t lin = blockIdx.x;
    if (NT && (gridDim.x & 7u) == 0u && !(opts & DIAG_OPT_PLAIN_ORDER)) lin = (lin & 7u) * (gridDim.x >> 3) + (lin >> 3);
    const uint32_t k = lin / n_slabs, slab = lin - k * n_slabs;
    const uint32_t b0 = slab * slab_rows;
    const uint32_t b1 = b0 + slab_rows < B ? b0 + slab_rows : B;
    const uint32_t tcol = threadIdx.x & (TPR - 1), rg = threadIdx.x / TPR;

    A w[CPT][VEC], bv[CPT][VEC];
    if (WANT_GX || relu_out) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            diag_w_chunk<T, LOG2D>(s1, s2, u, k, mean_plus, (uint32_t)(c * TPR + tcol) * VEC, w[c]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) bv[c][e] = (A)0;
            if (bias != nullptr) {
                const typename Chunk<T>::type bc = *reinterpret_cast<const typename Chunk<T>::type *>(bias + (c * TPR + tcol) * VEC);
#pragma unroll
                for (int e = 0; e < VEC; ++e) bv[c][e] = bc[e];
            }
        }
    }
    A acc[CPT][VEC], accb[CPT][VEC];
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[c][e] = accb[c][e] = (A)0;

    const u32x4 *gk = g + (int64_t)k * B * CPR;
    const u32x4 *xk = x + (XSHARED ? (int64_t)0 : (int64_t)k * B * CPR);
    u32x4 *gxk = WANT_GX ? gx + (int64_t)k * B * CPR : nullptr;
    for (uint32_t b = b0 + rg; b < b1; b += RG * UNR) {
        u32x4 rg_[UNR][CPT], rx_[UNR][CPT];
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const uint32_t row = b + i * RG;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                rg_[i][c] = rx_[i][c] = u32x4{0u, 0u, 0u, 0u};
                if (row < b1) {
                    const int64_t off = (int64_t)row * CPR + c * TPR + tcol;
                    rg_[i][c] = ld16<NT>(gk + off);
                    rx_[i][c] = ld16<NT && !XSHARED>(xk + off);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const uint32_t row = b + i * RG;
            if (row < b1) {
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    A gv[VEC], xv[VEC], xr[VEC];
                    E::unpack(rg_[i][c], gv);
                    E::unpack(rx_[i][c], xr);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) xv[e] = relu_in ? relu_(xr[e]) : xr[e];
                    if (relu_out) {
                        // the fused activation's backward: the forward's pre-activation x * w (+ bias) recomputed with the same
                        // roundings instead of a second 16-byte read per chunk; gradient passes unless that result is <= 0
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            A z = xv[e] * w[c][e];
                            if (bias != nullptr) z = z + bv[c][e];
                            if (relu_(z) <= (A)0) gv[e] = (A)0;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        acc[c][e] = acc[c][e] + gv[e] * xv[e];
                        accb[c][e] = accb[c][e] + gv[e];
                    }
                    if constexpr (WANT_GX) {
                        A o[VEC];
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            o[e] = gv[e] * w[c][e];
                            if (relu_in && xv[e] <= (A)0) o[e] = (A)0;
                        }
                        if (NT && !(opts & DIAG_OPT_PLAIN_STORES)) st16_stream(gxk + (int64_t)row * CPR + c * TPR + tcol, E::pack(o));
                        else st16<NT>(gxk + (int64_t)row * CPR + c * TPR + tcol, E::pack(o));
                    }
                }
            }
        }
    }
    T *p0 = part + ((size_t)(k * n_slabs + slab) * 2) * D;
    if constexpr (RG > 1) {
        // row groups -> one sum per column, in group order (CPT == 1 here)
        typedef typename Chunk<T>::type chunk_t;
        __shared__ __attribute__((aligned(16))) A red[2][256][VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { red[0][threadIdx.x][e] = acc[0][e]; red[1][threadIdx.x][e] = accb[0][e]; }
        __syncthreads();
        if (rg == 0) {
            for (int q = 1; q < RG; ++q)
#pragma unroll
                for (int e = 0; e < VEC; ++e) { acc[0][e] = acc[0][e] + red[0][q * TPR + tcol][e]; accb[0][e] = accb[0][e] + red[1][q * TPR + tcol][e]; }
            chunk_t o0, o1;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { o0[e] = acc[0][e]; o1[e] = accb[0][e]; }
            *reinterpret_cast<chunk_t *>(p0 + tcol * VEC) = o0;
            *reinterpret_cast<chunk_t *>(p0 + D + tcol * VEC) = o1;
        }
    } else {
        typedef typename Chunk<T>::type chunk_t;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            chunk_t o0, o1;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { o0[e] = acc[c][e]; o1[e] = accb[c][e]; }
            *reinterpret_cast<chunk_t *>(p0 + (c * TPR + tcol) * VEC) = o0;
            *reinterpret_cast<chunk_t *>(p0 + D + (c * TPR + tcol) * VEC) = o1;
        }
    }
}

// out: (4, U, D), U = mean_plus + S.  Row mean_plus + k of slot 0: dL/du_k; slot 1 / 2: sample k's share of dL/ds1, dL/ds2;
// slot 3: its share of dL/dbias.  (With the mean row the caller sums rows 1.. into row 0: dL/du_0 and the totals.)
template <typename T>
__global__ void __launch_bounds__(256)
diag_apply_bwd_finish_kernel(T *__restrict__ out, const T *__restrict__ part, const T *__restrict__ s1, const T *__restrict__ s2,
                             const T *__restrict__ u, uint32_t n_slabs, uint32_t D, uint32_t S, uint32_t mean_plus)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (i >= D) return;
    const T *p = part + (size_t)k * n_slabs * 2 * D + i;
    T gw[4] = {0, 0, 0, 0}, gb[4] = {0, 0, 0, 0};
    uint32_t s = 0;
    for (; s + 4 <= n_slabs; s += 4)
#pragma unroll
        for (int q = 0; q < 4; ++q) { gw[q] = gw[q] + p[(size_t)(s + q) * 2 * D]; gb[q] = gb[q] + p[(size_t)(s + q) * 2 * D + D]; }
    for (int q = 0; s < n_slabs; ++s, ++q) { gw[q] = gw[q] + p[(size_t)s * 2 * D]; gb[q] = gb[q] + p[(size_t)s * 2 * D + D]; }
    const T w = (gw[0] + gw[1]) + (gw[2] + gw[3]), bsum = (gb[0] + gb[1]) + (gb[2] + gb[3]);
    const T Dd = (T)D, a = s1[i], c = s2[i];
    const T uk = u[(size_t)(mean_plus + k) * D + i], u0 = mean_plus ? u[i] : (T)0;
    const uint32_t U = mean_plus + S;
    const size_t o = (size_t)(mean_plus + k) * D + i, slot = (size_t)U * D;
    out[o] = w * (a * Dd * c);
    out[slot + o] = w * (mean_plus ? Dd * (u0 * c) + Dd * (uk * c) : Dd * (uk * c));
    out[2 * slot + o] = w * (a * Dd * (mean_plus ? u0 + uk : uk));
    out[3 * slot + o] = bsum;
}

template <typename T> constexpr int diag_max_log2d() { return multi_pass_low_log2d<T>(); }      // 64-register tiles: f32 4096, f64 2048

inline int64_t diag_bwd_slabs(int64_t S, int64_t B, int rg)
{
    if (S < 1 || B < 1) return 1;
    const int64_t want = (8 * (int64_t)num_cu() + S - 1) / S;             // ~8 blocks per CU in total
    const int64_t most = (B + rg * 8 - 1) / (rg * 8);                     // at least 8 steps of rows per block
    int64_t n = want < most ? want : most;
    if (n < 1) n = 1;
    const int64_t slab_rows = (B + n - 1) / n;
    return (B + slab_rows - 1) / slab_rows;
}

template <typename T>
inline int diag_apply_dispatch(void *dst, const void *x, const void *s1, const void *s2, const void *u, const void *bias,
                               int64_t S, int64_t B, int32_t log2d, int32_t flags, void *stream)
{
    constexpr int LV = ilog2(Elem<T>::VEC);
    g_err[0] = 0;
    if (S < 0 || B < 0) return fail(WHVI_ERR_ARG, "whvi_diag_apply: negative size%s", "");
    if (flags & ~(WHVI_DIAG_X_SHARED | WHVI_DIAG_MEAN_PLUS | WHVI_DIAG_RELU_IN | WHVI_DIAG_RELU_OUT | WHVI_DIAG_TUNE_MASK))
        return fail(WHVI_ERR_ARG, "whvi_diag_apply: unknown flags%s 0x%llx", "", flags);
    if (log2d < LV || log2d > diag_max_log2d<T>())
        return fail(WHVI_ERR_SIZE, "whvi_diag_apply: log2(D)%s = %lld is outside the supported range [%lld, ...]", "", log2d, LV);
    const int64_t rows = S * B;
    if (rows == 0) return WHVI_OK;
    if (rows >= ((int64_t)1 << 32)) return fail(WHVI_ERR_SIZE, "whvi_diag_apply: rows are indexed with 32 bits%s", "");
    if (!dst || !x || !s1 || !s2 || !u) return fail(WHVI_ERR_ARG, "whvi_diag_apply: null pointer%s", "");
    if (((uintptr_t)dst | (uintptr_t)x | (uintptr_t)s1 | (uintptr_t)s2 | (uintptr_t)u | (uintptr_t)bias) & 15)
        return fail(WHVI_ERR_ALIGN, "whvi_diag_apply: a pointer%s is not 16-byte aligned", "");
    const bool shared = (flags & WHVI_DIAG_X_SHARED) != 0;
    {
        const char *d = (const char *)dst, *sp = (const char *)x;
        const int64_t dbytes = (rows << log2d) * (int64_t)sizeof(T), sbytes = ((shared ? B : rows) << log2d) * (int64_t)sizeof(T);
        if ((shared || d != sp) && d < sp + sbytes && sp < d + dbytes)
            return fail(WHVI_ERR_OVERLAP, "whvi_diag_apply: dst overlaps x%s", "");
    }
    hipStream_t st = (hipStream_t)stream;
    const uint32_t mean_plus = ((flags & WHVI_DIAG_MEAN_PLUS) ? DIAG_OPT_MEAN : 0u) | ((flags & WHVI_DIAG_RELU_IN) ? DIAG_OPT_RELU_IN : 0u) |
                               ((flags & WHVI_DIAG_RELU_OUT) ? DIAG_OPT_RELU_OUT : 0u);       // the kernels' option word
    const FastDiv db = make_fastdiv((uint32_t)B);
#define WHVI_DIAG(L, NT, SH) WHVI_DIAG_K(L, NT, SH, (pick_k<T, L>()))
#define WHVI_DIAG_K(L, NT, SH, KK)                                                                              \
    do {                                                                                                        \
        constexpr int K_ = KK;                                                                                  \
        const int64_t n_chunks = (rows << L) / Elem<T>::VEC, n_tiles = (n_chunks + 64 * K_ - 1) / (64 * K_);    \
        /* shared input: sample index fastest within an XCD when every sample is a whole number of 8-block groups */ \
        const int64_t blk_chunks = (int64_t)4 * 64 * K_, per_sample = (B << L) / Elem<T>::VEC;                  \
        /* (streaming launches only: at cache-resident sizes -- config 2's 256 MiB -- the plain order is faster, 54 vs 64 us) */ \
        const uint32_t fastest = (SH && NT && S > 1 && !(flags & WHVI_DIAG_TUNE_PLAIN_ORDER) && per_sample % (8 * blk_chunks) == 0) \
                                     ? (uint32_t)(per_sample / (8 * blk_chunks)) : 0u;                          \
        note_launch<T>("diag_apply_kernel", L, K_, (bool)NT, (bool)SH);                                         \
        hipLaunchKernelGGL((diag_apply_kernel<T, L, K_, NT, SH>), dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, \
                           (u32x4 *)dst, (const u32x4 *)x, (const T *)s1, (const T *)s2, (const T *)u, (const T *)bias, \
                           n_chunks, n_tiles, (uint32_t)rows, db, mean_plus, fastest);                          \
    } while (0)
#define WHVI_CASE(L)                                                                                            \
    case L:                                                                                                     \
        if constexpr (L >= LV && L <= diag_max_log2d<T>()) {                                                    \
            const bool nt = (flags & WHVI_DIAG_TUNE_NT) ? true : (flags & WHVI_DIAG_TUNE_CACHED) ? false           \
                            : stream_sized((rows << L) * (int64_t)sizeof(T), dst, shared ? nullptr : x);        \
            /* cache-resident results, rows of up to 256 chunks (f32 D <= 1024): quarter-size tiles -- four times the waves with a \
               quarter of the work each, as for the weight kernels (wbar_fwd.hpp): config 2 60 -> 50-53 us, D = 1024 +5 %, D = 64 \
               +9 %; NOT for one-row tiles of half the size (D = 2048: 34 -> 57 us).  WHVI_DIAG_TUNE_BIG_TILES: A/B */    \
            constexpr int NEED_ = (L > LV + 6) ? (1 << (L - LV - 6)) : 1;                                       \
            constexpr int KS_ = 4;                                                                              \
            const bool small_ = !nt && NEED_ <= KS_ && KS_ < pick_k<T, L>() && !(flags & WHVI_DIAG_TUNE_BIG_TILES); \
            if (small_) { if (shared) WHVI_DIAG_K(L, false, true, KS_); else WHVI_DIAG_K(L, false, false, KS_); } \
            else if (shared) { if (nt) WHVI_DIAG(L, true, true); else WHVI_DIAG(L, false, true); }              \
            else { if (nt) WHVI_DIAG(L, true, false); else WHVI_DIAG(L, false, false); }                        \
        }                                                                                                       \
        break;
    switch (log2d) {
        WHVI_CASE(1) WHVI_CASE(2) WHVI_CASE(3) WHVI_CASE(4) WHVI_CASE(5) WHVI_CASE(6) WHVI_CASE(7)
        WHVI_CASE(8) WHVI_CASE(9) WHVI_CASE(10) WHVI_CASE(11) WHVI_CASE(12)
    default: break;
    }
#undef WHVI_CASE
#undef WHVI_DIAG_K
#undef WHVI_DIAG
    return after_launch("diag_apply");
}

template <typename T>
inline int diag_apply_bwd_dispatch(void *grad_x, void *out, void *part, const void *g, const void *x, const void *s1,
                                   const void *s2, const void *u, const void *bias, int64_t S, int64_t B, int32_t log2d,
                                   int64_t n_slabs, int32_t flags, void *stream)
{
    constexpr int LV = ilog2(Elem<T>::VEC);
    g_err[0] = 0;
    if (S < 0 || B < 0) return fail(WHVI_ERR_ARG, "whvi_diag_apply_bwd: negative size%s", "");
    if (flags & ~(WHVI_DIAG_X_SHARED | WHVI_DIAG_MEAN_PLUS | WHVI_DIAG_RELU_IN | WHVI_DIAG_RELU_OUT | WHVI_DIAG_TUNE_MASK))
        return fail(WHVI_ERR_ARG, "whvi_diag_apply_bwd: unknown flags%s 0x%llx", "", flags);
    if (log2d < LV || log2d > diag_max_log2d<T>())
        return fail(WHVI_ERR_SIZE, "whvi_diag_apply_bwd: log2(D)%s = %lld is outside the supported range [%lld, ...]", "", log2d, LV);
    if (S == 0) return WHVI_OK;
    if (B == 0 || n_slabs < 1 || n_slabs > B || n_slabs > 65535 * 32 || S > 65535)
        return fail(WHVI_ERR_ARG, "whvi_diag_apply_bwd: bad batch / slab count%s (B = %lld, n_slabs = %lld)", "", B, n_slabs);
    if (S * B >= ((int64_t)1 << 32)) return fail(WHVI_ERR_SIZE, "whvi_diag_apply_bwd: rows are indexed with 32 bits%s", "");
    if (!out || !part || !g || !x || !s1 || !s2 || !u) return fail(WHVI_ERR_ARG, "whvi_diag_apply_bwd: null pointer%s", "");
    if (((uintptr_t)grad_x | (uintptr_t)out | (uintptr_t)part | (uintptr_t)g | (uintptr_t)x | (uintptr_t)s1 | (uintptr_t)s2 | (uintptr_t)u | (uintptr_t)bias) & 15)
        return fail(WHVI_ERR_ALIGN, "whvi_diag_apply_bwd: a pointer%s is not 16-byte aligned", "");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t mean_plus = (flags & WHVI_DIAG_MEAN_PLUS) ? 1u : 0u;
    // XCD-contiguous block order + write-through stores: measured (tools/diag_apply_rate.py, profiles/r04/diag_apply_rates.log)
    // +6.6 % on config 4's 9 GB (1.58 -> 1.48 ms, 6.06 TB/s) and -3 % on 2-3 GB streams (0.61 -> 0.63 ms): taken from 4 GiB up
    const bool long_stream = ((S * B) << log2d) * (int64_t)sizeof(T) * (grad_x ? 3 : 2) >= ((int64_t)4 << 30);
    const uint32_t opts = mean_plus | ((flags & WHVI_DIAG_RELU_IN) ? DIAG_OPT_RELU_IN : 0u) | ((flags & WHVI_DIAG_RELU_OUT) ? DIAG_OPT_RELU_OUT : 0u) |
                          (((flags & WHVI_DIAG_TUNE_PLAIN_ORDER) || !long_stream) ? (DIAG_OPT_PLAIN_ORDER | DIAG_OPT_PLAIN_STORES) : 0u);
    const bool shared = (flags & WHVI_DIAG_X_SHARED) != 0;
    const uint32_t slab_rows = (uint32_t)((B + n_slabs - 1) / n_slabs);
    if ((B + slab_rows - 1) / slab_rows != n_slabs)
        return fail(WHVI_ERR_ARG, "whvi_diag_apply_bwd: n_slabs%s = %lld is not a slab count of this batch (use whvi_diag_apply_bwd_slabs)", "", n_slabs);
    if (n_slabs * S >= ((int64_t)1 << 31)) return fail(WHVI_ERR_SIZE, "whvi_diag_apply_bwd: too many blocks%s", "");
    const dim3 grid((unsigned)(n_slabs * S));
#define WHVI_DBWD(L, NT, SH, GX)                                                                                \
    do {                                                                                                        \
        note_launch<T>("diag_apply_bwd_kernel", L, (bool)NT, (bool)SH, (bool)GX);                               \
        hipLaunchKernelGGL((diag_apply_bwd_kernel<T, L, NT, SH, GX>), grid, dim3(256), 0, st, (u32x4 *)grad_x, (T *)part, \
                           (const u32x4 *)g, (const u32x4 *)x, (const T *)s1, (const T *)s2, (const T *)u, (const T *)bias, \
                           (uint32_t)B, slab_rows, (uint32_t)n_slabs, opts);                                    \
    } while (0)
#define WHVI_CASE(L)                                                                                            \
    case L:                                                                                                     \
        if constexpr (L >= LV && L <= diag_max_log2d<T>()) {                                                    \
            const bool nt = (flags & WHVI_DIAG_TUNE_NT) ? true : (flags & WHVI_DIAG_TUNE_CACHED) ? false           \
                            : ((S * B) << L) * (int64_t)sizeof(T) * (grad_x ? 3 : 2) > NT_MIN_BYTES;            \
            if (grad_x) {                                                                                       \
                if (shared) { if (nt) WHVI_DBWD(L, true, true, true); else WHVI_DBWD(L, false, true, true); }   \
                else { if (nt) WHVI_DBWD(L, true, false, true); else WHVI_DBWD(L, false, false, true); }        \
            } else {                                                                                            \
                if (shared) { if (nt) WHVI_DBWD(L, true, true, false); else WHVI_DBWD(L, false, true, false); } \
                else { if (nt) WHVI_DBWD(L, true, false, false); else WHVI_DBWD(L, false, false, false); }      \
            }                                                                                                   \
        }                                                                                                       \
        break;
    switch (log2d) {
        WHVI_CASE(1) WHVI_CASE(2) WHVI_CASE(3) WHVI_CASE(4) WHVI_CASE(5) WHVI_CASE(6) WHVI_CASE(7)
        WHVI_CASE(8) WHVI_CASE(9) WHVI_CASE(10) WHVI_CASE(11) WHVI_CASE(12)
    default: break;
    }
#undef WHVI_CASE
#undef WHVI_DBWD
    int rc = after_launch("diag_apply_bwd");
    if (rc != WHVI_OK) return rc;
    const uint32_t D = 1u << log2d;
    hipLaunchKernelGGL((diag_apply_bwd_finish_kernel<T>), dim3((D + 255) / 256, (unsigned)S), dim3(256), 0, st, (T *)out,
                       (const T *)part, (const T *)s1, (const T *)s2, (const T *)u, (uint32_t)n_slabs, D, (uint32_t)S, mean_plus);
    return after_launch("diag_apply_bwd (finish)");
}

template <typename T>
inline int64_t diag_apply_bwd_slabs_for(int64_t S, int64_t B, int32_t log2d)
{
    constexpr int LV = ilog2(Elem<T>::VEC);
    if (log2d < LV || log2d > diag_max_log2d<T>()) return -1;
    const int cpr = 1 << (log2d - LV);
    return diag_bwd_slabs(S, B, cpr < 256 ? 256 / cpr : 1);
}

}  // namespace whvi
