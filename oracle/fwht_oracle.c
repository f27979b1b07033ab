/*
 * oracle/fwht_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's Walsh-Hadamard hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's library; the product (whvi_amd/) never imports it.
 *
 * Parity is PINNED: tests/test_oracle.py checks every function here against
 *   - the reference's own known-answer vectors (test/walsh.py:12-13,17-18),
 *   - the dense Hadamard identity the reference's tests use (test/walsh.py:22-49),
 *   - golden fixtures emitted by the live reference (tests/golden/make_golden.py),
 *   - and, when oracle/_ref is built, the reference C++ FWHT itself, bit for bit.
 *
 * Build:  make -C oracle   (gcc -O2 -ffp-contract=off -fwrapv; no fast-math, so every
 * add/sub/mul is one IEEE-754 rounding exactly as the reference's ATen ops, and
 * integer sums that overflow wrap modulo 2^32 / 2^64 as the reference's integer
 * tensors do -- pinned against oracle/_ref on overflowing inputs).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>

/*
 * Reference: src/fwht/cpp/fwht.cpp:7-18.  The reference transposes so that
 * x[j] is "element j of every row" and then, for h = 1, 2, 4, ... and every
 * pair (j, j+h) inside blocks of 2h:
 *     tmp = x[j] - x[j+h];  x[j] += x[j+h];  x[j+h] = tmp;
 * i.e. a radix-2 butterfly network in ASCENDING stride order, unnormalised,
 * natural (Sylvester) ordering.  Rows are independent; this is the same
 * loop nest applied to one contiguous row at a time.
 */
#define DEFINE_FWHT(NAME, T)                                                   \
    void NAME(T *x, int64_t rows, int64_t n)                                   \
    {                                                                          \
        for (int64_t r = 0; r < rows; ++r) {                                   \
            T *row = x + r * n;                                                \
            for (int64_t h = 1; h < n; h *= 2) {        /* fwht.cpp:8,17 */    \
                for (int64_t i = 0; i < n; i += 2 * h) { /* fwht.cpp:9 */      \
                    for (int64_t j = i; j < i + h; ++j) { /* fwht.cpp:10 */    \
                        T tmp = row[j] - row[j + h];     /* fwht.cpp:11 */     \
                        row[j] = row[j] + row[j + h];    /* fwht.cpp:12 */     \
                        row[j + h] = tmp;                /* fwht.cpp:13 */     \
                    }                                                          \
                }                                                              \
            }                                                                  \
        }                                                                      \
    }

DEFINE_FWHT(oracle_fwht_f32, float)
DEFINE_FWHT(oracle_fwht_f64, double)
DEFINE_FWHT(oracle_fwht_i32, int32_t)
DEFINE_FWHT(oracle_fwht_i64, int64_t)

/*
 * The reference's CUDA kernel (src/fwht/cuda/fwht_cuda_kernel.cu:94-138) runs
 * the same network with radix-4 groups from stride N/4 DOWN to 1 (plus one
 * radix-2 stage when log2 N is odd).  Stage order does not change integer
 * results but changes fp32 rounding; this variant exists so tests can state
 * the ascending-vs-descending tolerance (SURVEY.md finding 3) with numbers.
 * Inside one radix-4 group the reference does stride 2s first, then s
 * (fwht_cuda_kernel.cu:109-120: D0/D2 and D1/D3 pairs are i0/i2 = 2*stride
 * apart, then D0/D1 and D2/D3 = stride apart).
 */
#define DEFINE_FWHT_DESC(NAME, T)                                              \
    void NAME(T *x, int64_t rows, int64_t n)                                   \
    {                                                                          \
        for (int64_t r = 0; r < rows; ++r) {                                   \
            T *row = x + r * n;                                                \
            for (int64_t h = n / 2; h >= 1; h /= 2) {                          \
                for (int64_t i = 0; i < n; i += 2 * h) {                       \
                    for (int64_t j = i; j < i + h; ++j) {                      \
                        T a = row[j], b = row[j + h];                          \
                        row[j] = a + b;                                        \
                        row[j + h] = a - b;                                    \
                    }                                                          \
                }                                                              \
            }                                                                  \
        }                                                                      \
    }

DEFINE_FWHT_DESC(oracle_fwht_desc_f32, float)
DEFINE_FWHT_DESC(oracle_fwht_desc_f64, double)

/*
 * Scale / transform / scale / transform / scale pipeline
 *     y = a (.) FWHT( b (.) FWHT( c (.) x ) )
 * which is what WHVISquarePow2Matrix.w_bar / sample compute with x = diag(s2)
 * (src/weights.py:73,84), each "(.)" being one matmul_diag_left
 * (src/utils.py:4-12, row scaling) or matmul_diag_right (src/utils.py:15-23,
 * column scaling).  Every multiply is a separate rounding (the reference runs
 * them as separate ATen kernels), hence -ffp-contract=off.
 *
 * axis == 0  ("row", reference-exact dataflow of weights.py:73):
 *     scale vectors are indexed by the row's position inside its group:
 *     a[g*group_rows + r], r = row % group_rows, g = sample of the row.
 *     a is applied last (s1), b in the middle (u / g_tilde), c first; any of
 *     the three may be NULL (= all ones, no multiply performed).
 * axis == 1  ("col", textbook S1.H.diag(g).H.S2 applied to row vectors):
 *     scale vectors are indexed by the column j; a and c are shared (length
 *     n), b is per sample: b[s*n + j].
 *
 * Row r of the (rows, n) buffer belongs to sample s = (r / sample_stride) %
 * n_samples: sample_stride = 1 gives the (batch, sample, D) layout of
 * SURVEY.md 8(d); sample_stride = rows_per_sample gives (sample, row, D).
 */
void oracle_pipeline_f32(float *y, const float *x, const float *a,
                         const float *b, const float *c, int64_t rows,
                         int64_t n, int64_t n_samples, int64_t sample_stride,
                         int64_t group_rows, int axis)
{
    for (int64_t r = 0; r < rows; ++r) {
        const float *src = x + r * n;
        float *dst = y + r * n;
        int64_t s = (r / sample_stride) % n_samples;
        int64_t rr = r % group_rows;
        if (axis == 0) {
            float cs = c ? c[rr] : 1.0f;
            for (int64_t j = 0; j < n; ++j)
                dst[j] = c ? cs * src[j] : src[j];
            oracle_fwht_f32(dst, 1, n);
            if (b) {
                float bs = b[s * group_rows + rr];
                for (int64_t j = 0; j < n; ++j)
                    dst[j] = bs * dst[j];
            }
            oracle_fwht_f32(dst, 1, n);
            if (a) {
                float as = a[rr];
                for (int64_t j = 0; j < n; ++j)
                    dst[j] = as * dst[j];
            }
        } else {
            for (int64_t j = 0; j < n; ++j)
                dst[j] = c ? c[j] * src[j] : src[j];
            oracle_fwht_f32(dst, 1, n);
            if (b) {
                const float *bs = b + s * n;
                for (int64_t j = 0; j < n; ++j)
                    dst[j] = bs[j] * dst[j];
            }
            oracle_fwht_f32(dst, 1, n);
            if (a)
                for (int64_t j = 0; j < n; ++j)
                    dst[j] = a[j] * dst[j];
        }
    }
}

void oracle_pipeline_f64(double *y, const double *x, const double *a,
                         const double *b, const double *c, int64_t rows,
                         int64_t n, int64_t n_samples, int64_t sample_stride,
                         int64_t group_rows, int axis)
{
    for (int64_t r = 0; r < rows; ++r) {
        const double *src = x + r * n;
        double *dst = y + r * n;
        int64_t s = (r / sample_stride) % n_samples;
        int64_t rr = r % group_rows;
        for (int64_t j = 0; j < n; ++j) {
            double cv = c ? (axis == 0 ? c[rr] : c[j]) : 1.0;
            dst[j] = c ? cv * src[j] : src[j];
        }
        oracle_fwht_f64(dst, 1, n);
        if (b)
            for (int64_t j = 0; j < n; ++j)
                dst[j] = (axis == 0 ? b[s * group_rows + rr] : b[s * n + j]) * dst[j];
        oracle_fwht_f64(dst, 1, n);
        if (a)
            for (int64_t j = 0; j < n; ++j)
                dst[j] = (axis == 0 ? a[rr] : a[j]) * dst[j];
    }
}

/*
 * The same pipeline over batches of INDEPENDENT weight matrices: the sub-matrices
 * of WHVIStackedMatrix (src/weights.py:130-132,179-180) each carry their own s1 /
 * s2 / u, so the outer scale vectors are indexed by the row's sample as well --
 * flags bit 0: a is per sample, bit 1: c is per sample (b always is):
 *     a[s*group_rows + rr] (axis 0)      a[s*n + j] (axis 1), likewise c.
 * flags == 0 is oracle_pipeline_<type>.  One multiply = one rounding, as above.
 */
#define DEFINE_PIPELINE_EX(NAME, T, FWHT)                                      \
    void NAME(T *y, const T *x, const T *a, const T *b, const T *c,            \
              int64_t rows, int64_t n, int64_t n_samples,                      \
              int64_t sample_stride, int64_t group_rows, int axis, int flags)  \
    {                                                                          \
        const int64_t unit = axis == 0 ? group_rows : n;                       \
        for (int64_t r = 0; r < rows; ++r) {                                   \
            const T *src = x + r * n;                                          \
            T *dst = y + r * n;                                                \
            int64_t s = (r / sample_stride) % n_samples;                       \
            int64_t rr = r % group_rows;                                       \
            const T *av = a ? a + ((flags & 1) ? s * unit : 0) : 0;            \
            const T *bv = b ? b + s * unit : 0;                                \
            const T *cv = c ? c + ((flags & 2) ? s * unit : 0) : 0;            \
            for (int64_t j = 0; j < n; ++j)                                    \
                dst[j] = cv ? cv[axis == 0 ? rr : j] * src[j] : src[j];        \
            FWHT(dst, 1, n);                                                   \
            if (bv)                                                            \
                for (int64_t j = 0; j < n; ++j)                                \
                    dst[j] = bv[axis == 0 ? rr : j] * dst[j];                  \
            FWHT(dst, 1, n);                                                   \
            if (av)                                                            \
                for (int64_t j = 0; j < n; ++j)                                \
                    dst[j] = av[axis == 0 ? rr : j] * dst[j];                  \
        }                                                                      \
    }

DEFINE_PIPELINE_EX(oracle_pipeline_ex_f32, float, oracle_fwht_f32)
DEFINE_PIPELINE_EX(oracle_pipeline_ex_f64, double, oracle_fwht_f64)

/* Dense Sylvester-Hadamard entry, H[i][j] = (-1)^popcount(i & j): the closed
 * form of the recursion in src/utils.py:88-101 (build_H_recursive). */
int oracle_hadamard_entry(int64_t i, int64_t j)
{
    return (__builtin_popcountll((unsigned long long)(i & j)) & 1) ? -1 : 1;
}

/* y = (H @ x.T).T in double precision -- the independent check the
 * reference's tests use (test/walsh.py:26,46). O(rows * n^2): small n only. */
void oracle_dense_wht_f64(double *y, const double *x, int64_t rows, int64_t n)
{
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int64_t j = 0; j < n; ++j)
                acc += oracle_hadamard_entry(i, j) * x[r * n + j];
            y[r * n + i] = acc;
        }
}
