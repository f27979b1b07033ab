/*
 * include/whvi_hip.h -- C ABI of libwhvi_hip.so, the MI355X (gfx950) replacement for the
 * reference's native FWHT module.
 *
 * Drop-in boundary.  The reference binds ONE native entry point for this path:
 *
 *     at::Tensor fwht(at::Tensor X)            src/fwht/cuda/fwht_cuda.cpp:5-14,16-18
 *       -> fwht_cuda_frontend(X)               src/fwht/cuda/fwht_cuda_kernel.cu:156-181
 *          -> fwht_batch1_kernel / _batch2_    src/fwht/cuda/fwht_cuda_kernel.cu:35-146
 *
 * exported from a Python module named `fwht_cuda` (imported at src/fwht/cuda/fwht.py:2).
 * whvi_fwht_<dtype>() below is what that binding calls instead of fwht_cuda_frontend; the
 * Python shim `fwht_cuda.py` at the repo root re-creates the module on top of it (see
 * INTEGRATION.md).  The fused entry points replace the chains of ATen ops in
 * src/weights.py:73,84,92-93 (matmul_diag_left . fwht . matmul_diag_left . fwht).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  All buffers are caller-owned DEVICE
 *     pointers, 16-byte aligned, contiguous row-major (rows, 1 << log2d).  No ownership
 *     transfer, no allocation, no host synchronisation, no global mutable state other than
 *     a thread-local error string: safe to call from several host threads (the autograd
 *     engine calls backward from its own thread) and capturable in a hipGraph.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 *   - the kernel launches on the device that is current for the calling thread; the
 *     Python shim sets it from the tensor (the reference never does: fwht_cuda_kernel.cu:156).
 *   - dst == src is allowed everywhere (in-place); partial overlap is not.
 *   - return value 0 = launched; negative = WHVI_ERR_*; whvi_last_error() describes it.
 *     Unlike the reference (no cudaGetLastError, silent no-op for D < 4 or D > 4096,
 *     SURVEY.md 2a) every unsupported shape is an error and launch failures are reported.
 *
 * Transform definition: unnormalised Walsh-Hadamard transform in natural (Sylvester)
 * order of every row, y = x . H_D, computed as the radix-2 butterfly network in ASCENDING
 * stride order (h = 1, 2, 4, ...) with plain add/sub -- the order of src/fwht/cpp/fwht.cpp:7-18,
 * so f32/f64 results of whvi_fwht_<dtype> are bit-identical to that reference (sign of zero included, at every size) and
 * integer results are exact.  Two opt-in forms trade the sign of a ZERO result for speed and say so where they are declared:
 * whvi_fwht_ex with WHVI_FWHT_SIGNED_LANES, and the fused pipelines whvi_fused_shs_* (a result that is -0 in the reference's
 * arithmetic may come back as +0; every other bit, NaN positions included, is the reference's).
 */
#ifndef WHVI_HIP_H
#define WHVI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WHVI_HIP_ABI_VERSION 1

/* error codes */
#define WHVI_OK               0
#define WHVI_ERR_ARG         -1   /* null pointer, negative size, bad enum            */
#define WHVI_ERR_SIZE        -2   /* log2d outside [0, whvi_max_log2d(dtype)]          */
#define WHVI_ERR_ALIGN       -3   /* a pointer is not 16-byte aligned                  */
#define WHVI_ERR_LAUNCH      -4   /* hipGetLastError() after the launch was not success*/
#define WHVI_ERR_OVERLAP     -5   /* dst and src overlap without being equal           */

/* dtype enum used by the *_ex entry point and whvi_max_log2d */
#define WHVI_F32   0
#define WHVI_F64   1
#define WHVI_F16   2   /* IEEE half storage, f32 arithmetic, one rounding on store     */
#define WHVI_I32   3   /* two's complement, wraps like the reference's int tensors     */
#define WHVI_BF16  4   /* bfloat16 storage, f32 arithmetic, one rounding on store      */

int         whvi_hip_abi_version(void);
const char *whvi_last_error(void);          /* thread-local; "" when the last call succeeded */
int         whvi_last_kernel(char *buf, int32_t size);
                                            /* writes the demangled symbol of the kernel instantiation the calling
                                             * thread's last FWHT / fused launch selected (the name rocprofv3 prints,
                                             * e.g. "whvi::fwht_rows_kernel<float, 12, 16, 0, false, true, 256, 1>");
                                             * returns its length, 0 before any launch.  Measurement aid: the
                                             * reference has no counterpart (its launcher picks between two kernels
                                             * silently, fwht_cuda_kernel.cu:156-181)                          */
int         whvi_max_log2d(int32_t dtype);  /* largest supported log2(D): 24 for f32/f64/i32 (one pass up to
                                             * D = 65536, f64 32768: a wave per row, then a block per row;
                                             * longer rows take extra high-bit passes), 16 for f16/bf16
                                             * (single pass only: one rounding)                        */

/* Measurement aid (no counterpart in the reference): a copy of `bytes` bytes (a multiple of 16; dst == src allowed) with the
 * streaming geometry of the transform kernels and no arithmetic -- one 16 KiB tile per wave, 256-thread blocks,
 * XCD-contiguous block order, non-temporal loads, block barrier, write-through non-temporal stores.  Its rate is the
 * ceiling this memory system offers the access pattern whvi_fwht_* uses; bench.py prints it as roofline.ceiling_measured. */
int whvi_stream_copy_probe(void *dst, const void *src, int64_t bytes, void *stream);

/* Batched row FWHT: dst[r, :] = FWHT(src[r, :]) for r in [0, rows).
 * Replaces fwht_cuda_frontend (fwht_cuda_kernel.cu:156-181) + the X.clone() of
 * fwht_cuda.cpp:11 (pass dst != src for the reference's out-of-place semantics).
 *
 * Sign of zero: none to mention -- these entry points run the plain add / sub butterfly network at EVERY size, so f32 / f64
 * results carry the reference's bits including the sign of a zero result (a row of negative zeros gives [-0, +0, +0, ...],
 * as -0 + -0 = -0 does in src/fwht/cpp/fwht.cpp:11-13), whatever the launch form (cache-resident, streaming, block per row).
 * The faster lane-stage form that loses that sign is opt-in: WHVI_FWHT_SIGNED_LANES of whvi_fwht_ex below. */
int whvi_fwht_f32 (void *dst, const void *src, int64_t rows, int32_t log2d, void *stream);
int whvi_fwht_f64 (void *dst, const void *src, int64_t rows, int32_t log2d, void *stream);
int whvi_fwht_f16 (void *dst, const void *src, int64_t rows, int32_t log2d, void *stream);
int whvi_fwht_bf16(void *dst, const void *src, int64_t rows, int32_t log2d, void *stream);
int whvi_fwht_i32 (void *dst, const void *src, int64_t rows, int32_t log2d, void *stream);

/* Same transform with an explicit kernel variant, for tuning and for cross-checking the
 * cross-lane code paths against each other on hardware.  variant == 0 is the production launch
 * (what whvi_fwht_<dtype> does); otherwise
 *   bits 0..2 : butterfly network / loop form
 *                 0  DPP + permlane network, persistent loop with register prefetch of the next tile
 *                 1,5  every cross-lane stage through ds_bpermute (__shfl_xor), one tile per wave
 *                 2  DPP + permlane network, one tile per wave, cached accesses
 *                 3  LDS-staged network (fwht_tile_lds), one tile per wave, cached accesses
 *                 4  as 0 with non-temporal accesses
 *                 6  as 2 with non-temporal loads and write-through non-temporal stores (production form)
 *                 7  as 3 with non-temporal accesses
 *   bits 4..5 : threads per block, 0 = 256, 1 = 512, 2 = 1024 (576 for the LDS-staged network)
 *   bits 8..19: cap of the grid in blocks per CU (0 = uncapped: one tile per wave)
 *   bits 20..22: launch form of rows LONGER than one wavefront tile (D > 8192; f64 > 4096), every dtype:
 *                 0  production (a block per row up to D = 65536, pipelined grid chosen by size; pieces + passes beyond)
 *                 1  2^12-element pieces + high-bit passes for every long row (round 1's form)
 *                 2  one row per block      3  persistent pipelined grid (where that instantiation exists)
 *                 4  as 1 with every pass over the whole buffer instead of 128 MiB row groups
 *   bit 23 (WHVI_FWHT_SIGNED_LANES), with bits 0..19 zero: the production launch of whvi_fwht_<dtype>, except that STREAMING
 *                 launches (buffers beyond the 256 MiB Infinity Cache) of f32 rows of D = 512 .. 2048 and f64 rows of
 *                 D = 64 .. 2048 run their lane stages as fused multiply-adds by +/-1 (round 2-3: +1.7 %; since round 4's store
 *                 spacing the plain network streams as fast, 6.42-6.44 vs 6.43-6.45 TB/s, and this form is a cross-check of
 *                 the signed network the fused pipelines use).  A zero
 *                 carries no sign through those: a result that is NEGATIVE zero in the reference's arithmetic (element 0 of a
 *                 row of negative zeros) comes back as +0; every other bit is unchanged
 *                 (tests/test_fwht_gpu.py::test_negative_zero_contract_of_both_launch_forms).  Other shapes ignore the bit.
 * Variants other than the ds_bpermute cross-check exist for f32 and D = 512..4096 only; elsewhere only
 * bit 0 (and bits 20..23) are honoured.  All variants produce identical bits, WHVI_FWHT_SIGNED_LANES up to the sign of zero
 * (tests/test_fwht_gpu.py).
 * The library reads NO environment variables: its launch form is a function of its arguments.
 */
#define WHVI_FWHT_SIGNED_LANES (1 << 23)
int whvi_fwht_ex(void *dst, const void *src, int64_t rows, int32_t log2d,
                 int32_t dtype, int32_t variant, void *stream);

/* Fused scale -> FWHT -> scale -> FWHT -> scale, one HBM read + one write per row:
 *
 *     dst[r, :] = A (.) FWHT( B_s (.) FWHT( C (.) src[r, :] ) ),   s = sample of row r
 *
 * replacing matmul_diag_left(s1, fwht(matmul_diag_left(u, fwht(X)))) of src/weights.py:73,84.
 * Every multiply is a separate IEEE rounding (no FMA contraction), as in the reference's
 * separate ATen kernels.  Row r belongs to MC sample s = (r / sample_stride) % n_samples.
 *
 *   axis = WHVI_AXIS_ROW : the reference-exact dataflow (matmul_diag_left scales ROWS,
 *       src/utils.py:4-12).  Rows form groups of group_rows (= D for a D x D weight
 *       matrix); with i = r % group_rows:  A = a[i], B_s = b[s*group_rows + i], C = c[i]
 *       (per-row scalars).
 *   axis = WHVI_AXIS_COL : the textbook S1.H.diag(g).H.S2 applied to row vectors
 *       (matmul_diag_right, src/utils.py:15-23): A = a[j], B_s = b[s*D + j], C = c[j]
 *       for column j (group_rows is ignored).
 *   a, b, c may each be NULL (treated as all ones, the multiply is skipped).
 *   src may be NULL only with axis = WHVI_AXIS_ROW and group_rows <= D: the input is then the
 *       first group_rows rows of the D x D identity per group, i.e. with c = s2 and
 *       group_rows = D it is torch.diag(s2) of src/weights.py:73 without materialising it (no
 *       HBM read at all); group_rows = 1 yields row 0 only (all WHVIColumnMatrix uses,
 *       src/weights.py:245).
 */
#define WHVI_AXIS_ROW 0
#define WHVI_AXIS_COL 1

int whvi_fused_shs_f32(void *dst, const void *src, const void *a, const void *b,
                       const void *c, int64_t rows, int32_t log2d, int64_t n_samples,
                       int64_t sample_stride, int64_t group_rows, int32_t axis,
                       void *stream);
int whvi_fused_shs_f64(void *dst, const void *src, const void *a, const void *b,
                       const void *c, int64_t rows, int32_t log2d, int64_t n_samples,
                       int64_t sample_stride, int64_t group_rows, int32_t axis,
                       void *stream);

/* Same pipeline with per-sample outer scale vectors, for batches of INDEPENDENT weight matrices
 * (the sub-matrices of WHVIStackedMatrix, src/weights.py:130-132,179-180, or several MC samples of
 * several layers in one launch): with the flag set, a (resp. c) is indexed like b,
 *     a[s*group_rows + i]  (axis = ROW)      a[s*D + j]  (axis = COL),
 * i.e. every "sample" s carries its own s1 / s2 / u.  flags == 0 is whvi_fused_shs_<dtype>. */
#define WHVI_FUSED_A_PER_SAMPLE 1
#define WHVI_FUSED_C_PER_SAMPLE 2
/* WHVI_FUSED_SRC_SHARED (axis = COL, D * sizeof >= 1 KiB, dst != src): src holds the rows of ONE sample -- sample_stride rows
 * -- shared by all samples; row r of dst is computed from src row r % sample_stride.  With rows in (sample, batch, D)
 * order (sample_stride = batch) this is a layer's first Monte-Carlo pass on a (batch, D) input: the input is read from
 * the caches instead of being expanded to (n_samples, batch, D) in HBM first.  Same arithmetic per row: same bits as the
 * launch on the expanded input. */
#define WHVI_FUSED_SRC_SHARED   4
/* WHVI_FUSED_ONE_TRANSFORM (axis = COL, D * sizeof >= 1 KiB, c == NULL): dst = A (.) FWHT(B_s (.) src) -- the second half of
 * the pipeline on its own (matmul_diag . fwht, src/utils.py:15-23 + src/fwht).  For a source shared by all samples the
 * first half FWHT(C (.) x) is sample-independent: compute it once (this entry with b = C, n_samples = 1), then every
 * sample with WHVI_FUSED_SRC_SHARED | WHVI_FUSED_ONE_TRANSFORM -- one transform per sample instead of two, the same
 * multiplies and butterflies in the same order, hence the same bits as the two-transform launch. */
#define WHVI_FUSED_ONE_TRANSFORM 8

int whvi_fused_shs_ex_f32(void *dst, const void *src, const void *a, const void *b,
                          const void *c, int64_t rows, int32_t log2d, int64_t n_samples,
                          int64_t sample_stride, int64_t group_rows, int32_t axis,
                          int32_t flags, void *stream);
int whvi_fused_shs_ex_f64(void *dst, const void *src, const void *a, const void *b,
                          const void *c, int64_t rows, int32_t log2d, int64_t n_samples,
                          int64_t sample_stride, int64_t group_rows, int32_t axis,
                          int32_t flags, void *stream);

/* Reparameterisation + KL of J weight matrices in ONE launch, replacing the reference's
 * chain of small ATen kernels: g_sigma = softplus(g_rho) (src/weights.py:43-50), g_sigma * eps per MC sample
 * (src/weights.py:82-83,92), and kl_diag_normal(g_mu, g_sigma, 0, lambda) (src/weights.py:52-64,
 * src/utils.py:49-71, reference argument convention kept).  All buffers f32, contiguous:
 *   g_mu, g_rho : (J, D)        eps : (J, S, D)  drawn by the caller (injectable, graph-safe)
 *   u           : (J, 1+S, D)   u[j,0] = g_mu[j],  u[j,1+k] = sigma[j] * eps[j,k]   -> b of whvi_fused_shs_ex
 *   sigma       : (J, D)        kept for the backward pass
 *   kl_part     : (J, whvi_reparam_kl_blocks(D))  per-block partial sums; KL[j] = sum over the last axis
 */
int whvi_reparam_kl_blocks(int64_t D);
int whvi_reparam_kl_f32(void *u, void *sigma, void *kl_part, const void *g_mu, const void *g_rho,
                        const void *eps, int64_t J, int64_t S, int64_t D, float lambda_, void *stream);

/* The weight construction itself as a dedicated launch (src/weights.py:73; the generic route is
 * whvi_fused_shs_ex with src == NULL -- same bits):
 *     dst[j,k,i,:] = s1[j,i] * fwht( u[j,k,i] * fwht(s2[j,i] e_i) )  (+ base[j,i,:])
 * for the first R <= D rows i of each of the J x S matrices.  fwht(s2_i e_i) = s2_i H[i,:] is generated from bit
 * parities (the butterflies of a one-hot row are exact), so each row costs ONE transform, no HBM read, one write.
 * `base` (J, R, D), optional: a matrix added to every sample's matrix in the epilogue -- with base =
 * w_bar(g_mu) from a first call this is `w_bar(g_mu) + w_bar(g_sigma * eps_k)` of src/weights.py:93 without a
 * separate read-modify-write pass over all weight matrices.
 *   s1, s2 : (J, D)     dst : (J, S, R, D)     log2d in [2, 13] (f32) / [1, 12] (f64)
 *   u : (J, u_group, D); matrix (j, k) uses row u_first + k of group j.  u_group = S, u_first = 0 is the plain
 *       (J, S, D) layout; with the (J, 1 + S, D) buffer of whvi_reparam_kl the mean call passes S = 1, u_first = 0 and
 *       the per-sample call u_first = 1, both with u_group = 1 + S_samples -- no slicing copies. */
int whvi_wbar_fwd_f32(void *dst, const void *s1, const void *u, const void *s2, const void *base,
                      int64_t J, int64_t S, int64_t R, int32_t log2d, int64_t u_group, int64_t u_first,
                      void *stream);
int whvi_wbar_fwd_f64(void *dst, const void *s1, const void *u, const void *s2, const void *base,
                      int64_t J, int64_t S, int64_t R, int32_t log2d, int64_t u_group, int64_t u_first,
                      void *stream);

/* `w_bar(g_mu) + w_bar(g_sigma * eps_k)` of src/weights.py:93 in ONE launch: u is (J, 1 + S, D) -- row 0 of each group the
 * mean vector, rows 1 .. S the samples (the buffer whvi_reparam_kl writes) -- and
 *     dst[j,k,i,:] = s1[j,i] * fwht(u[j,0,i] * fwht(s2[j,i] e_i))  +  s1[j,i] * fwht(u[j,1+k,i] * fwht(s2[j,i] e_i)),
 * both terms computed by the same wave (two in-register transforms per tile, no mean matrix in memory).  The same
 * multiplies, butterflies and final add as whvi_wbar_fwd twice (mean matrix, then samples with `base`): the same bits.
 * Meant for cache-resident results (a launch and the re-read of the mean matrix saved); streams keep the two-launch
 * form, which does half the arithmetic per byte written.   dst : (J, S, R, D); log2d in [2, 12] (f32) / [1, 11] (f64):
 * two 64-register tiles per wave -- longer rows return WHVI_ERR_SIZE (use whvi_wbar_fwd with `base`). */
int whvi_wbar_fwd_mean_f32(void *dst, const void *s1, const void *u, const void *s2, int64_t J, int64_t S, int64_t R,
                           int32_t log2d, void *stream);
int whvi_wbar_fwd_mean_f64(void *dst, const void *s1, const void *u, const void *s2, int64_t J, int64_t S, int64_t R,
                           int32_t log2d, void *stream);

/* Backward of the weight construction in ONE launch: reads the incoming gradient once, writes three scalars per
 * row.  The reference obtains the same quantities from autograd over its op chain (matmul_diag_left backward,
 * src/utils.py:4-12, and FWHTFunction.backward = FWHT, src/fwht/cuda/fwht.py:14-16): four more FWHT launches and
 * ~10 elementwise / reduction launches over (J, S, R, D) tensors.  Buffers of the entry point's dtype, contiguous:
 *   grad_w  : (J, S, R, D)  dL/dW, first R <= D rows of every matrix      s1, s2 : (J, D)      u : (J, S, D)
 *   grad_u  : (J, S, D)     dL/du[j,k,i] for i < R; entries i >= R are NOT written (zero-fill them when R < D)
 *   part_s1 : (J, S, D)     per-sample contributions, same convention; dL/ds1[j,i] = sum over k (likewise part_s2)
 * flags = WHVI_WBAR_MEAN: W[j,k] = w_bar(u[j,0]) + w_bar(u[j,1+k]) (the whvi_wbar_fwd `base` form): u and the three
 *   outputs are (J, 1 + S, D); rows 1..S are written (part_s1 / part_s2 include the mean vector's share) and row 0 is
 *   left to the caller: the sum over k of rows 1..S is dL/du[j,0] resp. the per-matrix totals.
 * log2d in [2, 13] (f32) / [1, 12] (f64). */
#define WHVI_WBAR_MEAN   1
#define WHVI_WBAR_NO_LDS 2   /* tuning / cross-check: butterflies through the DPP network instead of the LDS-staged one
                              * (same adds in the same order: identical transform bits) */
#define WHVI_WBAR_SMALL_TILES 4   /* tuning / cross-check: force the quarter-size tiles small problems take ... */
#define WHVI_WBAR_BIG_TILES   8   /* ... or forbid them (default: chosen by size) */
int whvi_wbar_bwd_f32(void *grad_u, void *part_s1, void *part_s2, const void *grad_w, const void *s1,
                      const void *u, const void *s2, int64_t J, int64_t S, int64_t R, int32_t log2d,
                      int32_t flags, void *stream);
int whvi_wbar_bwd_f64(void *grad_u, void *part_s1, void *part_s2, const void *grad_w, const void *s1,
                      const void *u, const void *s2, int64_t J, int64_t S, int64_t R, int32_t log2d,
                      int32_t flags, void *stream);

/* `h @ (w_bar(g_mu) + w_bar(g_sigma * eps_k)).T (+ bias)` of src/weights.py:87-93,101-102 for all MC samples in ONE launch,
 * without building the matrices.  As written in the reference w_bar(u) = S1 . fwht(diag(u) . fwht(diag(s2))) is EXACTLY
 * D * diag(s1 (.) u (.) s2) -- row scales around row transforms, SURVEY.md finding 1 -- so the dense product adds exact zeros
 * to one product per output.  This entry computes that product directly, with the roundings of the as-written chain in the
 * same order ( v = u_i * s2_i;  D * v exact;  s1_i * .;  mean + sample;  h * w;  + bias ):
 *     out[k, b, i] = x[(k,) b, i] * ( wd(u[0])_i + wd(u[1 + k])_i ) + bias[i],     wd(u)_i = s1_i * (D * (u_i * s2_i)),
 * identical to the matrix route (weight construction + GEMM) for every input -- bit for bit against rocBLAS on MI355X, zeros
 * included: the product is added to +0, the value the other D - 1 products (exact zeros) sum to in a GEMM's +0-initialised
 * accumulator, so a product of -0 comes out as +0 like there.  Non-finite operands are propagated as the matrix
 * route propagates them: a row of x holding an inf / NaN at column j makes every other output of that row NaN (inf * 0 in
 * the dot product), a non-finite s1_i or an overflowing partial sum of the second transform (|D/2 * u_i * s2_i| = inf) makes
 * output column i NaN.
 *   x    : (S, B, D), or (B, D) with WHVI_DIAG_X_SHARED (a layer's first pass: one input for all samples; dst must not overlap)
 *   u    : (1 + S, D) with WHVI_DIAG_MEAN_PLUS (row 0 = g_mu, row 1 + k = g_sigma * eps_k: the buffer whvi_reparam_kl
 *          writes), else (S, D) and w_k = wd(u[k]) alone (direct weight sampling, src/weights.py:75-85,104-108)
 *   s1, s2 : (D,)    bias : (D,) or NULL    out : (S, B, D); out == x is allowed without WHVI_DIAG_X_SHARED
 *   log2d in [2, 12] (f32) / [1, 11] (f64). */
#define WHVI_DIAG_X_SHARED  1
#define WHVI_DIAG_MEAN_PLUS 2
/* Element-wise neighbours fused in (an nn.ReLU in front of / behind the layer in the caller's nn.Sequential, src/networks.py:49
 * running the reference's module list): x is passed through max(x, 0) on load, resp. the result before the store -- the
 * values torch.relu gives (NaN stays NaN), one read + one write of the activations instead of three. */
#define WHVI_DIAG_RELU_IN   4
#define WHVI_DIAG_RELU_OUT  8
/* tuning / cross-check (same values either way): force the streaming (non-temporal) or the cached launch instead of the
 * choice by size; keep the plain block order on a shared input */
#define WHVI_DIAG_TUNE_NT          16
#define WHVI_DIAG_TUNE_CACHED      32
#define WHVI_DIAG_TUNE_PLAIN_ORDER 64
#define WHVI_DIAG_TUNE_BIG_TILES   128   /* 16 KiB tiles at cache-resident sizes too (default there: quarter-size tiles) */
#define WHVI_DIAG_TUNE_MASK        (16 | 32 | 64 | 128)
int whvi_diag_apply_f32(void *out, const void *x, const void *s1, const void *s2, const void *u, const void *bias,
                        int64_t S, int64_t B, int32_t log2d, int32_t flags, void *stream);
int whvi_diag_apply_f64(void *out, const void *x, const void *s1, const void *s2, const void *u, const void *bias,
                        int64_t S, int64_t B, int32_t log2d, int32_t flags, void *stream);

/* Backward of whvi_diag_apply (closed form; replaces autograd over the GEMM, the weight construction and its op chain):
 *   grad_x : (S, B, D) = g[k, b, :] (.) w_k, or NULL when the input needs no gradient (with WHVI_DIAG_X_SHARED the caller
 *            sums it over k)
 *   out    : (4, U, D), U = S (+ 1 with WHVI_DIAG_MEAN_PLUS); row r = k (+ 1) of slot 0 = dL/du[r], of slots 1 / 2 = sample
 *            k's share of dL/ds1 / dL/ds2, of slot 3 = its share of dL/dbias = sum_b g[k, b, :].  With the mean row, row 0
 *            is left to the caller: the sum of rows 1 .. S is dL/du[0] resp. the totals (one reduction for all four).
 *   part   : workspace of S * n_slabs * 2 * D elements, n_slabs = whvi_diag_apply_bwd_slabs(dtype, S, B, log2d) -- the
 *            batch reduction sum_b g (.) x runs over n_slabs row slabs per sample, combined in slab order by a second,
 *            tiny launch inside the same call (deterministic summation order; no atomics).
 * Non-finite g propagates element-wise (no attempt to mimic the matrix route on a diverged backward pass).
 * x is the forward's input as it was passed (BEFORE a fused WHVI_DIAG_RELU_IN); bias (the forward's, or NULL) is read only with
 * WHVI_DIAG_RELU_OUT, whose mask is recomputed from x, the diagonal and the bias with the forward's roundings (torch's
 * threshold_backward: the gradient passes unless the activation's result is <= 0). */
int64_t whvi_diag_apply_bwd_slabs(int32_t dtype, int64_t S, int64_t B, int32_t log2d);
int whvi_diag_apply_bwd_f32(void *grad_x, void *out, void *part, const void *g, const void *x, const void *s1,
                            const void *s2, const void *u, const void *bias, int64_t S, int64_t B, int32_t log2d,
                            int64_t n_slabs, int32_t flags, void *stream);
int whvi_diag_apply_bwd_f64(void *grad_x, void *out, void *part, const void *g, const void *x, const void *s1,
                            const void *s2, const void *u, const void *bias, int64_t S, int64_t B, int32_t log2d,
                            int64_t n_slabs, int32_t flags, void *stream);

/* The two dense products either side of the square layer in a WHVI regression network, for all Monte-Carlo samples per launch,
 * as HBM-bound streams (f32; same dataflow as the reference's matrix products -- every product is formed, exact zeros of the
 * as-written weights included, so non-finite inputs propagate like in the dense product):
 *   whvi_small_k_apply_f32 : out[s, b, n] = sum_{c < K} x[b, c] * w[s, n, c] (+ bias[n]) (relu with WHVI_APPLY_RELU_OUT)
 *       `x_padded @ W.T` of WHVIStackedMatrix with a narrow input (src/weights.py:179-180,195-206; WHVILinear(3, 1024): K = 4,
 *       N = 1024).  x : (B, K) shared by all samples; w : (S, N, K); out : (S, B, N); K = 2^log2k in {4, 8}; N a multiple of 4
 *       with N / 4 = (a power of two <= 256) x (1 .. 4).  Fused multiply-adds in ascending c.
 *   whvi_row_dot_f32 : y[s, b] = sum_i x[s, b, i] * w[s, i] (+ bias[0])  (x through max(., 0) first with WHVI_APPLY_RELU_IN)
 *       `F.linear(x, w[None], bias)` of the transposed WHVIColumnMatrix (src/weights.py:239-251; WHVILinear(1024, 1)).
 *       x : (S, B, D); w : (S, D); y : (S, B); log2d in [2, 12].  Per-lane partial sums in ascending column order, then a
 *       butterfly over the wave's lanes: a different summation order than a GEMV's (same tolerance class). */
#define WHVI_APPLY_RELU_IN  1
#define WHVI_APPLY_RELU_OUT 2
int whvi_small_k_apply_f32(void *out, const void *x, const void *w, const void *bias, int64_t S, int64_t B, int64_t N,
                           int32_t log2k, int32_t flags, void *stream);
int whvi_row_dot_f32(void *y, const void *x, const void *w, const void *bias, int64_t S, int64_t B, int32_t log2d,
                     int32_t flags, void *stream);

/* whvi_reparam_kl_f32 with the eps draw inside the kernel: Philox4x32-10 + Box-Muller, one standard
 * normal per (matrix, sample, element), written to eps_out (J, S, D) for the backward pass / inspection.  The
 * generator state is three 64-bit words in DEVICE memory, state = {seed, launch offset, scratch (must be 0)}; the
 * kernel advances the offset itself when its last block finishes, so nothing about the draw is baked into the
 * launch: safe to capture in a hipGraph (every replay draws fresh eps).  One stream per state at a time.  The
 * stream of numbers is this library's own (not torch.randn's); for bit-level parity work inject eps through
 * whvi_reparam_kl_f32, which computes the same u / sigma / KL from a given eps. */
int whvi_reparam_kl_philox_f32(void *u, void *sigma, void *kl_part, void *eps_out, const void *g_mu,
                               const void *g_rho, void *state, int64_t J, int64_t S, int64_t D, float lambda_,
                               void *stream);

/* Backward of whvi_reparam_kl_f32 in one launch (closed form): grad_u (J, 1+S, D) and grad_kl (J) are the incoming
 * gradients (either may be NULL = zero), sigma the forward's saved output; writes grad_mu, grad_rho (J, D).
 * Replaces autograd over softplus / mul / kl_diag_normal (src/weights.py:43-64,82-83; src/utils.py:49-71). */
int whvi_reparam_kl_bwd_f32(void *grad_mu, void *grad_rho, const void *grad_u, const void *grad_kl,
                            const void *g_mu, const void *g_rho, const void *eps, const void *sigma,
                            int64_t J, int64_t S, int64_t D, float lambda_, void *stream);

/* Gaussian mean negative log likelihood of MC predictions, replacing the per-output Python loop and
 * the elementwise chain of GaussianLikelihood.mnll_batch_estimate (src/likelihoods.py:18-29) by one reduction:
 *     part[b] = { scale * sum_b log N(y | y_hat, sigma^2),  sum_b z^2 },   z = (y - y_hat) / sigma
 * over three index dimensions size[0..2] (HOST arrays) with element strides yhat_stride / y_stride (y broadcasts
 * with stride 0 along the MC axis; order the dimensions by decreasing y_hat stride for coalesced reads);
 * mnll = sum_b part[b][0] with scale = -n / (m * n_mc); b < whvi_gauss_mnll_blocks(size[0]*size[1]*size[2]).
 * sigma is a DEVICE scalar (the learnable likelihood.sigma).  The backward writes
 *     grad_yhat = g * scale * (y - y_hat) / sigma^2   (same strides as y_hat),
 *     grad_sigma = g * scale * (sum z^2 - count) / sigma        with g = *grad_out (device scalar). */
int whvi_gauss_mnll_blocks(int64_t total);
int whvi_gauss_mnll_f32(void *part, const void *y, const void *y_hat, const void *sigma, const int64_t *size,
                        const int64_t *yhat_stride, const int64_t *y_stride, float scale, void *stream);
int whvi_gauss_mnll_bwd_f32(void *grad_yhat, void *grad_sigma, const void *grad_out, const void *part,
                            const void *y, const void *y_hat, const void *sigma, const int64_t *size,
                            const int64_t *yhat_stride, const int64_t *y_stride, float scale, void *stream);

/* Learning-rate schedule of the reference's experiments on the device (src/evaluation.py:25-26: LambdaLR with
 * lambda t: lambda0 * (1 + gamma * t) ** (-p); src/networks.py:80-81 steps it after every batch):
 *     if (advance) *t += 1;   *lr = (float)(base_lr * (lambda0 * pow(1 + gamma * *t, -p)))
 * t: DEVICE double (the step counter), lr: DEVICE float (the rate a capturable Adam reads).  One single-thread launch,
 * capture-safe; float64 arithmetic like LambdaLR's Python floats, one rounding to float32.  With several parameter
 * groups call it once per group, advance = 1 on the first only. */
int whvi_decay_lr_step(void *t, void *lr, double base_lr, double lambda0, double gamma, double p, int advance,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WHVI_HIP_H */
