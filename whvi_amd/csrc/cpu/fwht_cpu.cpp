// whvi_amd/csrc/cpu/fwht_cpu.cpp -- native CPU FWHT for HOST tensors (libwhvi_cpu.so).
//
// Product replacement for the reference's `fwht_cpp` extension (src/fwht/cpp/fwht.cpp:3-34),
// which spends its time in three ATen dispatches per butterfly pair on stride-D views.  Same
// result bit for bit (ascending strides 1, 2, 4, ..., one add or sub per output), but each row
// is transformed in place in contiguous memory, rows are spread over the host cores with
// OpenMP, and the pair loops are written so the compiler can vectorise them.
//
// This library only ever sees HOST pointers: GPU tensors go to libwhvi_hip.so and nowhere else.
#include <stdint.h>
#include <string.h>

namespace {

template <typename T>
inline void row_transform(T *__restrict__ v, int64_t n)
{
    // strides 1 and 2 together on groups of four (still the ascending order: stride 1 first)
    if (n >= 4) {
        for (int64_t i = 0; i < n; i += 4) {
            T a = v[i] + v[i + 1], b = v[i] - v[i + 1];
            T c = v[i + 2] + v[i + 3], d = v[i + 2] - v[i + 3];
            v[i] = a + c;
            v[i + 1] = b + d;
            v[i + 2] = a - c;
            v[i + 3] = b - d;
        }
    } else if (n == 2) {
        T a = v[0] + v[1], b = v[0] - v[1];
        v[0] = a;
        v[1] = b;
        return;
    } else {
        return;
    }
    for (int64_t h = 4; h < n; h <<= 1) {
        for (int64_t blk = 0; blk < n; blk += 2 * h) {
            T *__restrict__ lo = v + blk;
            T *__restrict__ hi = v + blk + h;
#pragma omp simd
            for (int64_t j = 0; j < h; ++j) {
                T x = lo[j], y = hi[j];
                lo[j] = x + y;
                hi[j] = x - y;
            }
        }
    }
}

template <typename T>
inline void rows_transform(T *dst, const T *src, int64_t rows, int64_t n)
{
#pragma omp parallel for schedule(static) if (rows * n >= 16384)
    for (int64_t r = 0; r < rows; ++r) {
        T *out = dst + r * n;
        if (dst != src) memcpy(out, src + r * n, sizeof(T) * (size_t)n);
        row_transform(out, n);
    }
}

inline bool bad(const void *dst, const void *src, int64_t rows, int64_t n)
{
    return dst == nullptr || src == nullptr || rows < 0 || n < 1 || (n & (n - 1)) != 0;
}

}  // namespace

#define WHVI_CPU_API extern "C" __attribute__((visibility("default")))

// dst[r, :] = FWHT(src[r, :]); dst == src allowed.  Returns 0, or -1 on a bad argument.
WHVI_CPU_API int whvi_cpu_fwht_f32(float *dst, const float *src, int64_t rows, int64_t n)
{
    if (bad(dst, src, rows, n)) return -1;
    rows_transform(dst, src, rows, n);
    return 0;
}
WHVI_CPU_API int whvi_cpu_fwht_f64(double *dst, const double *src, int64_t rows, int64_t n)
{
    if (bad(dst, src, rows, n)) return -1;
    rows_transform(dst, src, rows, n);
    return 0;
}
WHVI_CPU_API int whvi_cpu_fwht_i32(uint32_t *dst, const uint32_t *src, int64_t rows, int64_t n)
{
    if (bad(dst, src, rows, n)) return -1;
    rows_transform(dst, src, rows, n);  // unsigned: wraps like the reference's int tensors
    return 0;
}
WHVI_CPU_API int whvi_cpu_fwht_i64(uint64_t *dst, const uint64_t *src, int64_t rows, int64_t n)
{
    if (bad(dst, src, rows, n)) return -1;
    rows_transform(dst, src, rows, n);
    return 0;
}
WHVI_CPU_API int whvi_cpu_abi_version(void) { return 1; }
