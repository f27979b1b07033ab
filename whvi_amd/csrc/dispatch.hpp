#pragma once
// whvi_amd/csrc/dispatch.hpp -- host-side argument checks and launch selection shared by the
// per-dtype translation units (fwht_<dtype>.hip, fused_<dtype>.hip).  The C ABI itself is
// declared in include/whvi_hip.h.
#include "kernels.hpp"

namespace whvi {

// thread-local error text returned by whvi_last_error(); defined in abi.hip
extern thread_local char g_err[512];

int fail(int code, const char *fmt, const char *a = "", long long x = 0, long long y = 0);
int after_launch(const char *what);
int num_cu();

template <typename T> constexpr int max_log2d()
{
    // log2(VEC) + 6 lane bits + log2(Kmax): f32/i32 2+6+5 (K=32) = 13; f16/bf16 3+6+4 (K=16) = 13;
    // f64 1+6+5 (K=32) = 12.  Beyond that a row no longer fits one wave's registers.
    return sizeof(T) == 8 ? 12 : 13;
}

// K = 16-byte chunks per lane: the smallest power of two that holds one row, but never below
// the streaming size (64 lanes * 16 chunks * 16 B = 16 KiB of loads in flight per wave).
template <typename T, int LOG2D> constexpr int pick_k()
{
    constexpr int LV = ilog2(Elem<T>::VEC);
    constexpr int need = (LOG2D > LV + 6) ? (1 << (LOG2D - LV - 6)) : 1;
    constexpr int stream = 16;
    return need > stream ? need : stream;
}

// data registers per lane held by one tile (32-bit units)
template <typename T, int K> constexpr int tile_vgprs()
{
    return K * Elem<T>::VEC * (int)sizeof(typename Elem<T>::acc) / 4;
}

inline int check_common(const void *dst, const void *src, int64_t rows, int32_t log2d, int maxl,
                        size_t elem, bool src_optional = false)
{
    g_err[0] = 0;
    if (rows == 0 && log2d >= 0 && log2d <= maxl) return WHVI_OK;   // nothing to do, pointers may be null
    if (dst == nullptr || (src == nullptr && !src_optional))
        return fail(WHVI_ERR_ARG, "whvi: null %s pointer", dst ? "src" : "dst");
    if (rows < 0) return fail(WHVI_ERR_ARG, "whvi: negative row count%s (%lld)", "", rows);
    if (log2d < 0 || log2d > maxl)
        return fail(WHVI_ERR_SIZE, "whvi: log2(D)%s = %lld is outside the supported range [0, %lld]", "",
                    log2d, maxl);
    if (((uintptr_t)dst & 15) || ((uintptr_t)src & 15))
        return fail(WHVI_ERR_ALIGN, "whvi: %s pointer is not 16-byte aligned",
                    ((uintptr_t)dst & 15) ? "dst" : "src");
    if (src != nullptr && dst != src) {
        const char *d = (const char *)dst, *s = (const char *)src;
        const int64_t bytes = (rows << log2d) * (int64_t)elem;
        if (d < s + bytes && s < d + bytes)
            return fail(WHVI_ERR_OVERLAP, "whvi: dst and src overlap without being equal%s", "");
    }
    return WHVI_OK;
}

// variant word of whvi_fwht_ex (include/whvi_hip.h): bit0 shfl, bit1 no-prefetch, bit2 non-temporal,
// bits 8..19 blocks per CU of the persistent grid.  FULL = every variant is compiled (tuning
// sizes); otherwise only bit0 is honoured.
template <typename T, int LOG2D, int K, bool FULL>
inline void launch_rows(void *dst, const void *src, int64_t n_chunks, int variant, hipStream_t st)
{
    constexpr bool CAN_PREFETCH = tile_vgprs<T, K>() <= 64;
    const int64_t n_tiles = (n_chunks + 64 * K - 1) / (64 * K);
    int bpc = (variant >> 8) & 0xFFF;
    if (bpc == 0) bpc = CAN_PREFETCH ? 3 : 2;
    const int64_t want = (n_tiles + 3) / 4;
    const int64_t cap = (int64_t)num_cu() * bpc;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    u32x4 *d = (u32x4 *)dst;
    const u32x4 *s = (const u32x4 *)src;
#define WHVI_LAUNCH(POL, PF, NT)                                                                      \
    hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POL, PF, NT>), dim3(grid), dim3(256), 0, st, d, \
                       s, n_chunks, n_tiles)
    if constexpr (FULL && CAN_PREFETCH) {
        switch (variant & 7) {
        case 0: WHVI_LAUNCH(POLICY_DPP, true, false); break;
        case 1: WHVI_LAUNCH(POLICY_SHFL, true, false); break;
        case 2: WHVI_LAUNCH(POLICY_DPP, false, false); break;
        case 3: WHVI_LAUNCH(POLICY_SHFL, false, false); break;
        case 4: WHVI_LAUNCH(POLICY_DPP, true, true); break;
        case 5: WHVI_LAUNCH(POLICY_SHFL, true, true); break;
        case 6: WHVI_LAUNCH(POLICY_DPP, false, true); break;
        default: WHVI_LAUNCH(POLICY_SHFL, false, true); break;
        }
    } else {
        if (variant & 1) WHVI_LAUNCH(POLICY_SHFL, CAN_PREFETCH, false);
        else WHVI_LAUNCH(POLICY_DPP, CAN_PREFETCH, false);
    }
#undef WHVI_LAUNCH
}

template <typename T, bool TUNABLE>
inline int fwht_dispatch(void *dst, const void *src, int64_t rows, int32_t log2d, int variant, void *stream)
{
    int rc = check_common(dst, src, rows, log2d, max_log2d<T>(), sizeof(T));
    if (rc != WHVI_OK) return rc;
    if (rows == 0) return WHVI_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t elems = rows << log2d;
    constexpr int VEC = Elem<T>::VEC;
    const int64_t n_chunks = elems / VEC;   // whole 16-byte chunks
    const int64_t tail_elems = elems - n_chunks * VEC;

#define WHVI_CASE(L)                                                                                 \
    case L:                                                                                          \
        if constexpr (L <= max_log2d<T>()) {                                                         \
            if (n_chunks > 0)                                                                        \
                launch_rows<T, L, pick_k<T, L>(), (TUNABLE && L >= 9 && L <= 12)>(dst, src, n_chunks, \
                                                                                  variant, st);      \
            if constexpr ((1 << L) < VEC) {                                                          \
                if (tail_elems > 0) {                                                                \
                    const int64_t first = (n_chunks * VEC) >> L;                                     \
                    const int64_t n = rows - first;                                                  \
                    hipLaunchKernelGGL((fwht_tail_kernel<T, L>), dim3((unsigned)((n + 63) / 64)),     \
                                       dim3(64), 0, st, (T *)dst, (const T *)src, first, rows);       \
                }                                                                                    \
            }                                                                                        \
        }                                                                                            \
        break;
    switch (log2d) {
        WHVI_CASE(0) WHVI_CASE(1) WHVI_CASE(2) WHVI_CASE(3) WHVI_CASE(4) WHVI_CASE(5) WHVI_CASE(6)
        WHVI_CASE(7) WHVI_CASE(8) WHVI_CASE(9) WHVI_CASE(10) WHVI_CASE(11) WHVI_CASE(12) WHVI_CASE(13)
    default: break;
    }
#undef WHVI_CASE
    return after_launch("fwht");
}

// ---- fused pipeline --------------------------------------------------------------------------
template <typename T, int LOG2D>
inline void launch_fused(void *dst, const void *src, const void *a, const void *b, const void *c,
                         int64_t rows, int64_t n_samples, int64_t sample_stride, int64_t group_rows,
                         int axis, hipStream_t st)
{
    constexpr int K = pick_k<T, LOG2D>();
    constexpr int VEC = Elem<T>::VEC;
    const int64_t n_chunks = (rows << LOG2D) / VEC;
    const int64_t n_tiles = (n_chunks + 64 * K - 1) / (64 * K);
    const int64_t want = (n_tiles + 3) / 4;
    const int64_t cap = (int64_t)num_cu() * (tile_vgprs<T, K>() <= 64 ? 4 : 2);
    const unsigned grid = (unsigned)(want < cap ? want : cap);
#define WHVI_FUSED(AX, EYE)                                                                         \
    hipLaunchKernelGGL((fused_shs_kernel<T, LOG2D, K, AX, EYE>), dim3(grid), dim3(256), 0, st,       \
                       (u32x4 *)dst, (const u32x4 *)src, (const T *)a, (const T *)b, (const T *)c,  \
                       n_chunks, n_tiles, n_samples, sample_stride, group_rows)
    if (src == nullptr) WHVI_FUSED(WHVI_AXIS_ROW, true);
    else if (axis == WHVI_AXIS_ROW) WHVI_FUSED(WHVI_AXIS_ROW, false);
    else WHVI_FUSED(WHVI_AXIS_COL, false);
#undef WHVI_FUSED
}

template <typename T>
inline int fused_dispatch(void *dst, const void *src, const void *a, const void *b, const void *c,
                          int64_t rows, int32_t log2d, int64_t n_samples, int64_t sample_stride,
                          int64_t group_rows, int32_t axis, void *stream)
{
    constexpr int LV = ilog2(Elem<T>::VEC);
    int rc = check_common(dst, src, rows, log2d, max_log2d<T>(), sizeof(T), true);
    if (rc != WHVI_OK) return rc;
    if (axis != WHVI_AXIS_ROW && axis != WHVI_AXIS_COL)
        return fail(WHVI_ERR_ARG, "whvi: bad axis%s %lld", "", axis);
    if (n_samples < 1 || sample_stride < 1 || group_rows < 1)
        return fail(WHVI_ERR_ARG, "whvi: n_samples, sample_stride and group_rows must be >= 1%s", "");
    if (log2d < LV)
        return fail(WHVI_ERR_SIZE, "whvi: the fused pipeline needs D >= %s%lld elements (one 16-byte chunk)",
                    "", (long long)Elem<T>::VEC);
    if (src == nullptr && (axis != WHVI_AXIS_ROW || group_rows != ((int64_t)1 << log2d)))
        return fail(WHVI_ERR_ARG, "whvi: src == NULL (identity input) needs axis = ROW and group_rows == D%s", "");
    if (axis == WHVI_AXIS_COL && (((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)c & 15)))
        return fail(WHVI_ERR_ALIGN, "whvi: column scale vectors must be 16-byte aligned%s", "");
    if (rows == 0) return WHVI_OK;
    hipStream_t st = (hipStream_t)stream;
#define WHVI_CASE(L)                                                                                     \
    case L:                                                                                              \
        if constexpr (L >= LV && L <= max_log2d<T>())                                                    \
            launch_fused<T, L>(dst, src, a, b, c, rows, n_samples, sample_stride, group_rows, axis, st); \
        break;
    switch (log2d) {
        WHVI_CASE(1) WHVI_CASE(2) WHVI_CASE(3) WHVI_CASE(4) WHVI_CASE(5) WHVI_CASE(6) WHVI_CASE(7)
        WHVI_CASE(8) WHVI_CASE(9) WHVI_CASE(10) WHVI_CASE(11) WHVI_CASE(12) WHVI_CASE(13)
    default: break;
    }
#undef WHVI_CASE
    return after_launch("fused_shs");
}

}  // namespace whvi
