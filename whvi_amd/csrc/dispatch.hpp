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

// Which kernel instantiation the calling thread's last launch selected -- the demangled symbol rocprofv3 prints,
// formatted lazily by whvi_last_kernel() (abi.hip).  A few stores per launch; lets bench.py name the kernel it
// actually timed instead of a literal.
struct LaunchNote {
    const char *family = nullptr;     // "fwht_rows_kernel", "fused_shs_kernel", ...
    const char *type = nullptr;
    int n = 0;                        // template arguments after the type
    int arg[12] = {0};
    unsigned char is_bool[12] = {0};
};
extern thread_local LaunchNote g_note;
template <typename T> constexpr const char *type_name()
{
    if (std::is_same<T, float>::value) return "float";
    if (std::is_same<T, double>::value) return "double";
    if (std::is_same<T, int32_t>::value) return "int";
    if (std::is_same<T, __half>::value) return "__half";
    return "__hip_bfloat16";
}
struct NB { int v; bool b; };                       // template argument + "print as bool"
inline NB nb(int v) { return NB{v, false}; }
inline NB nb(bool v) { return NB{v ? 1 : 0, true}; }
template <typename T, typename... A>
inline void note_launch(const char *family, A... a)
{
    const NB args[] = {nb(a)...};
    g_note.family = family;
    g_note.type = type_name<T>();
    static_assert(sizeof...(A) <= 12, "LaunchNote holds twelve template arguments");
    g_note.n = (int)sizeof...(A);
    for (int i = 0; i < (int)sizeof...(A) && i < 12; ++i) {
        g_note.arg[i] = args[i].v;
        g_note.is_bool[i] = args[i].b;
    }
}

// MC samples one block of the reparameterisation kernels covers (abi.hip, train_aux.hip)
constexpr int REPARAM_SAMPLES_PER_BLOCK = 8;

// Largest row one wavefront holds in registers: log2(VEC) + 6 lane bits + log2(Kmax): f32/i32 2+6+5 (K=32) = 13;
// f16/bf16 3+6+4 (K=16) = 13; f64 1+6+5 (K=32) = 12.  Longer rows take the row kernel on 2^LOW-element pieces
// plus high-bit passes (fwht_high_kernel).
template <typename T> constexpr int max_single_pass_log2d() { return sizeof(T) == 8 ? 12 : 13; }
template <typename T> constexpr int multi_pass_low_log2d() { return sizeof(T) == 8 ? 11 : 12; }   // 64-VGPR tiles
// One block per row (fwht_block_rows_kernel): 2 .. 16 wave tiles of 2^LOW elements, so rows up to 2^(LOW + 4) take
// one pass as well: f32 / i32 / f16 / bf16 D <= 65536, f64 D <= 32768.
template <typename T> constexpr int max_block_log2d() { return multi_pass_low_log2d<T>() + 4; }
// 16-bit storage types stay single-pass: their contract is ONE rounding of the f32 result, and a pass boundary
// would round the intermediate too.
template <typename T> constexpr int max_log2d() { return sizeof(T) == 2 ? max_block_log2d<T>() : 24; }

// K = 16-byte chunks per lane: the smallest power of two that holds one row, but never below
// the streaming size -- as many chunks as keep the tile at 64 data VGPRs (f32/i32/f64: 16 chunks =
// 16 KiB per wave, f16/bf16: 8 chunks = 8 KiB), so the tile still fits a 1024-thread block.
template <typename T, int LOG2D> constexpr int pick_k()
{
    constexpr int LV = ilog2(Elem<T>::VEC);
    constexpr int need = (LOG2D > LV + 6) ? (1 << (LOG2D - LV - 6)) : 1;
    constexpr int words = Elem<T>::VEC * (int)sizeof(typename Elem<T>::acc) / 4;   // VGPRs per chunk
    constexpr int stream = 64 / words;   // (16 KiB tiles for f16 were tried: 128 data VGPRs, 4.55 vs 5.5 TB/s)
    return need > stream ? need : stream;
}

// data registers per lane held by one tile (32-bit units)
template <typename T, int K> constexpr int tile_vgprs()
{
    return K * Elem<T>::VEC * (int)sizeof(typename Elem<T>::acc) / 4;
}

// Threads per block of the streaming launch.  1024 (16 waves = 256 KiB contiguous per block) is the
// best memory geometry but caps the kernel at 128 VGPRs; instantiations that would spill at that
// cap use 512 (tools/check_spills.py fails the build if any shipped kernel uses scratch).
template <typename T, int LOG2D> constexpr int big_block()
{
    if (tile_vgprs<T, pick_k<T, LOG2D>()>() > 64) return 256;
    if (std::is_same<T, float>::value) return (LOG2D >= 7 || LOG2D <= 2) ? 1024 : 512;
    if (std::is_same<T, int32_t>::value) return (LOG2D >= 7 || LOG2D <= 2) ? 1024 : 512;
    if (std::is_same<T, double>::value) return (LOG2D >= 6) ? 1024 : 512;
    if (std::is_same<T, __hip_bfloat16>::value) return (LOG2D >= 8 || LOG2D <= 3) ? 1024 : 512;
    return 512;   // __half: the pack/unpack temporaries do not fit 128 VGPRs
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

// Launch geometry (measured on MI355X, tools/membench.hip + tools/probe_variants.py):
// the HBM system rewards (a) one tile per wave and out -- no persistent loop, no register prefetch,
// (b) 1024-thread blocks, so 16 waves that start together cover 256 KiB contiguous, and (c)
// non-temporal loads/stores for streams far larger than the 256 MiB Infinity Cache.  Small problems
// use 256-thread blocks (more CUs busy) and cached accesses (the consumer is usually next in line).
constexpr int64_t NT_MIN_BYTES = (int64_t)256 << 20;
// Streaming (non-temporal) launch iff the launch's working set -- the buffer once when in place, source plus
// destination otherwise -- EXCEEDS the 256 MiB Infinity Cache.  Measured crossover (tools/probe_nt_threshold.py,
// D = 4096 f32): in place, cached accesses win up to 256 MiB (6.7 vs 5.7 TB/s) and lose from 320 MiB (5.7 vs 5.8);
// out of place they win up to 128 MiB per buffer (7.3 vs 5.4) and lose from 192 MiB (5.2 vs 5.7).
inline bool stream_sized(int64_t bytes, const void *dst, const void *src)
{
    return (dst == src || src == nullptr ? bytes : 2 * bytes) > NT_MIN_BYTES;
}

// Streaming launch of the f32 one-row tile of 128 data registers (D = 8192) at THREE waves per SIMD.  Defined -- with its
// kernel -- in fwht_wide.hip, a translation unit compiled with -fno-slp-vectorize: at 168 VGPRs the even-aligned register
// pairs of the packed adds the SLP vectoriser forms are what does not fit (with them: 28-120 B / lane of scratch; without:
// none).  The 64-register tiles WANT those pairs (6.46 vs 6.30 TB/s, 5.1), so they stay in fwht_f32.hip.
template <typename T>
void launch_wide_stream(u32x4 *d, const u32x4 *s, int64_t n_chunks, int64_t n_tiles, hipStream_t st);

// variant word of whvi_fwht_ex: documented in include/whvi_hip.h (0 = production launch).
// FULL = the extra tuning variants are compiled (f32, D = 512..4096).
// signed_lanes (whvi_fwht_ex variant bit 23, WHVI_FWHT_SIGNED_LANES): the production launch, but streams of f32 rows of
// 512 .. 2048 and f64 rows of 64 .. 2048 run their lane stages as signed fused multiply-adds (+1.7 %; a NEGATIVE zero result
// comes back as +0).  whvi_fwht_<dtype> never sets it: its bits are the reference's at every size.
template <typename T, int LOG2D, int K, bool FULL>
inline void launch_rows(void *dst, const void *src, int64_t n_chunks, int variant, hipStream_t st, bool signed_lanes = false)
{
    constexpr bool SMALL_TILE = tile_vgprs<T, K>() <= 64;   // fits 1024-thread blocks / prefetch
    const int64_t n_tiles = (n_chunks + 64 * K - 1) / (64 * K);
    u32x4 *d = (u32x4 *)dst;
    const u32x4 *s = (const u32x4 *)src;
    const int bpc = (variant >> 8) & 0xFFF;
    constexpr size_t rows_slab_bytes = (size_t)K * (64 * Elem<T>::VEC + Elem<T>::VEC) * 4;
#define WHVI_LAUNCH(POL, PF, NT, BLK)                                                                  \
    do {                                                                                               \
        int64_t grid = (n_tiles + (BLK / 64) - 1) / (BLK / 64);                                        \
        if (bpc > 0 && grid > (int64_t)num_cu() * bpc) grid = (int64_t)num_cu() * bpc;                 \
        const size_t smem = (POL == POLICY_LDS) ? (size_t)(BLK / 64) * rows_slab_bytes : 0;          \
        note_launch<T>("fwht_rows_kernel", LOG2D, K, (int)POL, (bool)PF, (bool)NT, (int)BLK, 0, false); \
        hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POL, PF, NT, BLK>), dim3((unsigned)grid),    \
                           dim3(BLK), smem, st, d, s, n_chunks, n_tiles);                              \
    } while (0)
    if (variant == 0) {   // production path
        const bool big = n_tiles >= (int64_t)32 * num_cu();
        const bool nt = stream_sized(n_chunks * 16, dst, src);
        constexpr int BIG = big_block<T, LOG2D>();
        if constexpr (BIG > 256) {
            if (big && nt) {
                // streams: 256-thread blocks (one wave per SIMD: the four waves run in step) with a block
                // barrier in front of the stores, so each block writes its 64 KiB back together -- measured
                // 6.33 vs 6.24 TB/s for 1024-thread blocks without the barrier (with it, 1024-thread blocks
                // fall to 5.7: their waves share SIMDs and leave the butterflies microseconds apart); fp16 5.5 ->
                // 6.2, bf16 5.0 -> 6.1, i32 5.8 -> 6.3 TB/s (tools/probe_stream_blocks.py)
                static const bool exp_big_blocks = WHVI_TUNE_ENV("WHVI_STREAM_BIG_BLOCKS") != nullptr;   // A/B switch (tuning builds)
                // f64 (64-register tiles, D = 64 .. 2048): 256-thread blocks + store barrier with the SIGNED DPP network (one
                // fma per lane-stage element instead of a sign fold + an add, one repair multiply at the end): 6.35-6.38 vs
                // 6.25-6.28 TB/s for the round-2 launch (1024-thread blocks, no barrier), which in turn beats 256 + barrier
                // with the UNSIGNED network (5.93-5.97) -- gpurun_out r03_ab_f64form*.log; D < 64 keeps the 512-thread form
                // Round 4: the UNSIGNED network in the same 256-thread + barrier launch streams at 6.39-6.41 TB/s since its
                // stores are issued one slot apart (kernels.hpp: SINGLE_PASS / STORE_NOP) -- what the signed form gave
                // (6.40) -- so the drop-in entry point takes that and keeps the sign of zero; the signed network is opt-in
                // (signed_lanes), 1024-thread blocks without the barrier (6.03-6.07) only for rows shorter than 64.
                // profiles/r04/stream_forms_store_spacing_ab.log
                constexpr bool F64_SIGNED = sizeof(T) == 8 && LOG2D >= 6 && WHVI_F64_STREAM_FORM == 2;
                if (exp_big_blocks || (sizeof(T) == 8 && (LOG2D < 6 || WHVI_F64_STREAM_FORM == 0))) WHVI_LAUNCH(POLICY_DPP, false, true, BIG);
                else if (!signed_lanes && sizeof(T) == 8) {
                    note_launch<T>("fwht_rows_kernel", LOG2D, K, POLICY_DPP, false, true, 256, 1, false);
                    hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POLICY_DPP, false, true, 256, 1, false>),
                                       dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, d, s, n_chunks, n_tiles);
                }
                else if constexpr (F64_SIGNED) {
                    note_launch<T>("fwht_rows_kernel", LOG2D, K, POLICY_DPP, false, true, 256, 1, true);
                    hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POLICY_DPP, false, true, 256, 1, true>),
                                       dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, d, s, n_chunks, n_tiles);
                }
                else if constexpr (sizeof(T) == 2 && K * Elem<T>::VEC == 64)
                    // 16-bit storage: half the bytes per butterfly, so the DPP network's VALU time co-limits the
                    // stream (6.1 TB/s).  The LDS-staged network needs a third of the issue slots: fp16 6.4,
                    // bf16 6.5 TB/s (tools/probe_f16.py), even at 8 waves per CU (16.6 KB of LDS per wave).
                {
                    note_launch<T>("fwht_rows_kernel", LOG2D, K, POLICY_LDS, false, true, 256, 1, false);
                    hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POLICY_LDS, false, true, 256, 1>),
                                       dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), (size_t)4 * rows_slab_bytes, st,
                                       d, s, n_chunks, n_tiles);
                }
                else {
                    // f32 D = 512 .. 2048 with signed_lanes: the signed DPP network (+1.7 %, kernels.hpp); everything else unsigned
                    constexpr bool SG = std::is_same<T, float>::value && LOG2D >= 9 && LOG2D <= 11;
                    if (SG && signed_lanes) {
                        note_launch<T>("fwht_rows_kernel", LOG2D, K, POLICY_DPP, false, true, 256, 1, SG);
                        hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POLICY_DPP, false, true, 256, 1, SG>),
                                           dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, d, s, n_chunks, n_tiles);
                    } else {
                        note_launch<T>("fwht_rows_kernel", LOG2D, K, POLICY_DPP, false, true, 256, 1, false);
                        hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POLICY_DPP, false, true, 256, 1, false>),
                                           dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, d, s, n_chunks, n_tiles);
                    }
                }
            } else if (big && sizeof(T) == 8) WHVI_LAUNCH(POLICY_DPP, false, false, BIG);
            // cache-resident sizes: 256-thread blocks match or beat 1024 at every size (tools/probe_midsize.py:
            // 6.6-6.8 vs 6.5-6.7 TB/s at 128-256 MiB in place, 6.7 vs 5.3 at 32 MiB)
            else WHVI_LAUNCH(POLICY_DPP, false, false, 256);
        } else {
            // tiles of more than 64 data VGPRs (one row per wave: f32 D = 8192, f64 D = 4096): 256-thread blocks
            // either way; streams get the non-temporal accesses and the store barrier as well
            if (big && nt) {
                constexpr bool SG = (sizeof(T) == 8 || std::is_same<T, float>::value) && WHVI_F64_STREAM_FORM == 3;   // tuning: the signed network here loses (6.23 vs 6.31)
                if constexpr (!SG && std::is_same<T, float>::value && LOG2D == max_single_pass_log2d<T>() && WHVI_WIDE_TILE_WAVES > 0)
                    launch_wide_stream<T>(d, s, n_chunks, n_tiles, st);
                else {
                    note_launch<T>("fwht_rows_kernel", LOG2D, K, POLICY_DPP, false, true, 256, 1, SG);
                    hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POLICY_DPP, false, true, 256, 1, SG>),
                                       dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, d, s, n_chunks, n_tiles);
                }
            } else WHVI_LAUNCH(POLICY_DPP, false, false, 256);
        }
        return;
    }
    if constexpr (FULL && SMALL_TILE) {
        const int blk = (variant >> 4) & 3;
        switch (variant & 7) {
        case 0: WHVI_LAUNCH(POLICY_DPP, true, false, 256); break;
        case 1: WHVI_LAUNCH(POLICY_SHFL, false, false, 256); break;
        case 2:
            if (blk == 0) WHVI_LAUNCH(POLICY_DPP, false, false, 256);
            else if (blk == 1) WHVI_LAUNCH(POLICY_DPP, false, false, 512);
            else WHVI_LAUNCH(POLICY_DPP, false, false, 1024);
            break;
        case 3:
            if constexpr (sizeof(typename Elem<T>::acc) == 4 && K * Elem<T>::VEC == 64) {
                if (blk == 0) WHVI_LAUNCH(POLICY_LDS, false, false, 256);
                else if (blk == 1) WHVI_LAUNCH(POLICY_LDS, false, false, 512);
                else WHVI_LAUNCH(POLICY_LDS, false, false, 576);
            }
            break;
        case 7:
            if constexpr (sizeof(typename Elem<T>::acc) == 4 && K * Elem<T>::VEC == 64) {
                if (blk == 0 && ((variant >> 6) & 3) != 0 && bpc == 0)
                    hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POLICY_LDS, false, true, 256, 1>),
                                       dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), (size_t)4 * rows_slab_bytes, st,
                                       d, s, n_chunks, n_tiles);
                else if (blk == 0) WHVI_LAUNCH(POLICY_LDS, false, true, 256);
                else if (blk == 1) WHVI_LAUNCH(POLICY_LDS, false, true, 512);
                else WHVI_LAUNCH(POLICY_LDS, false, true, 576);
            }
            break;
        case 4: WHVI_LAUNCH(POLICY_DPP, true, true, 256); break;
        case 6: {
            const int align = (variant >> 6) & 3;   // bits 6..7: store-alignment experiment (uncapped grids only)
#define WHVI_LAUNCH_A(BLK, AL)                                                                             \
    hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POLICY_DPP, false, true, BLK, AL>),                  \
                       dim3((unsigned)((n_tiles + (BLK / 64) - 1) / (BLK / 64))), dim3(BLK), 0, st, d, s, n_chunks, n_tiles)
            if (align == 1 && bpc > 0 && blk == 0 && (n_tiles & 3) == 0) {
                // persistent experiment: bpc blocks per CU walk the tiles with a grid stride, store barrier inside
                // the loop (whole blocks share a trip count because n_tiles is a multiple of the 4 waves per block)
                int64_t grid = (int64_t)num_cu() * bpc;
                if (grid > n_tiles / 4) grid = n_tiles / 4;
                grid &= ~(int64_t)7;                  // multiple of 8: the XCD-contiguous order stays a bijection
                if (grid < 8) grid = 8;
                note_launch<T>("fwht_rows_kernel", LOG2D, K, POLICY_DPP, false, true, 256, 1, false);
                hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POLICY_DPP, false, true, 256, 1>), dim3((unsigned)grid),
                                   dim3(256), 0, st, d, s, n_chunks, n_tiles);
            } else if (align == 0 || bpc > 0) {
                if (blk == 0) WHVI_LAUNCH(POLICY_DPP, false, true, 256);
                else if (blk == 1) WHVI_LAUNCH(POLICY_DPP, false, true, 512);
                else if (blk == 2) WHVI_LAUNCH(POLICY_DPP, false, true, 1024);
                else WHVI_LAUNCH(POLICY_DPP, false, true, 128);
            } else if (align == 1) {
                if (blk == 0) WHVI_LAUNCH_A(256, 1);
                else if (blk == 1) WHVI_LAUNCH_A(512, 1);
                else if (blk == 2) WHVI_LAUNCH_A(1024, 1);
                else WHVI_LAUNCH_A(128, 1);
            } else {
                if (blk == 0) WHVI_LAUNCH_A(256, 2);
                else if (blk == 1) WHVI_LAUNCH_A(512, 2);
                else if (blk == 2) WHVI_LAUNCH_A(1024, 2);
                else WHVI_LAUNCH_A(128, 2);
            }
#undef WHVI_LAUNCH_A
            break;
        }
        default: WHVI_LAUNCH(POLICY_SHFL, false, false, 256); break;
        }
    } else {
        if (variant & 1) WHVI_LAUNCH(POLICY_SHFL, false, false, 256);
        else WHVI_LAUNCH(POLICY_DPP, false, false, 256);
    }
#undef WHVI_LAUNCH
}

// One block of 2^LOG2W waves per row of 2^(LOW + LOG2W) elements (kernels.hpp: fwht_block_rows_kernel).
// form (whvi_fwht_ex variant bits 20..22, cross-checks): 0 = chosen by size, LONG_ONE_ROW = one row per block, LONG_PIPE =
// the persistent pipelined grid (where the instantiation exists and every resident block has a row)
constexpr int LONG_PASSES = 1, LONG_ONE_ROW = 2, LONG_PIPE = 3, LONG_PASSES_UNGROUPED = 4;
template <typename T, int LOG2W>
inline void launch_block_rows(void *dst, const void *src, int64_t n_rows, int form, hipStream_t st)
{
    constexpr int W = 1 << LOG2W;
    constexpr size_t smem = (size_t)W * 8 * 1024;
    constexpr int LOW = multi_pass_low_log2d<T>();
    const bool nt = stream_sized((n_rows << (LOW + LOG2W)) * (int64_t)sizeof(T), dst, src);
#ifdef WHVI_BLOCK_TRACE
    // probe builds: WHVI_BLOCK_TRACE=<device pointer, hex> receives 8 x uint64 per row (tools/probe_block_trace.py)
    static const uintptr_t trace_env = [] { const char *e = WHVI_TUNE_ENV("WHVI_BLOCK_TRACE"); return e ? (uintptr_t)strtoull(e, nullptr, 16) : (uintptr_t)0; }();
    uint64_t *trace = (uint64_t *)trace_env;
#else
    uint64_t *trace = nullptr;
#endif
    // Persistent, software-pipelined grid (kernels.hpp) once every resident block has several rows to walk; measured on
    // 4 GiB in place (profiles/r02/long_rows_pipe_ab.log): 16-wave blocks f32 4.89 -> 5.95, i32 5.71 -> 6.01 TB/s; 4-wave
    // blocks f32 5.68 -> 5.95, f64 5.71 -> 5.89; 8-wave blocks lose 2 % (two blocks per CU already overlap each other);
    // 16-bit storage loses 20 % at 4 / 8 waves and gains nothing at 16 (half the bytes per butterfly: those rows are bound
    // by the exchange, not by HBM), so it has no pipelined instantiations; neither has f64 at 16 waves, which needs 10
    // registers more than a 1024-thread block has.  `form` (whvi_fwht_ex) forces the choice where both forms exist (A/B).
    const int pipe_env = form == LONG_ONE_ROW ? 0 : (form == LONG_PIPE ? 1 : -1);
    const int resident = num_cu() * (16 / W);                        // blocks the chip holds at once (16 waves per CU)
    constexpr bool HAS_PIPE = sizeof(T) == 4 || (sizeof(T) == 8 && W < 16);
    const bool pipe = HAS_PIPE && (pipe_env >= 0 ? (pipe_env != 0 && n_rows > resident) : (W != 8 && n_rows >= 4 * (int64_t)resident));
    const int64_t grid = pipe ? resident : n_rows;                   // n_rows <= 2^31 - 1: far beyond any buffer
    note_launch<T>("fwht_block_rows_kernel", LOG2W, nt, pipe);
#define WHVI_BLOCK_ROWS(NTV, PIPEV)                                                                              \
    do {                                                                                                         \
        if constexpr (smem > 64 * 1024)   /* per launch, not once: the attribute belongs to the CURRENT device's copy */ \
            (void)hipFuncSetAttribute((const void *)fwht_block_rows_kernel<T, LOG2W, NTV, PIPEV>,                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                    \
        hipLaunchKernelGGL((fwht_block_rows_kernel<T, LOG2W, NTV, PIPEV>), dim3((unsigned)grid), dim3(64 * W), smem, st, \
                           (u32x4 *)dst, (const u32x4 *)src, n_rows, trace);                                     \
    } while (0)
    if constexpr (HAS_PIPE) {
        if (nt && pipe) { WHVI_BLOCK_ROWS(true, true); return; }
        if (pipe) { WHVI_BLOCK_ROWS(false, true); return; }
    }
    if (nt) WHVI_BLOCK_ROWS(true, false);
    else WHVI_BLOCK_ROWS(false, false);
#undef WHVI_BLOCK_ROWS
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
    // variant bits 20..22 (whvi_fwht_ex): launch form of rows longer than one wave tile, for cross-checks and A/Bs --
    // LONG_PASSES: the round-1 path (2^LOW-element pieces + high-bit passes) for every long row
    const int long_form = (variant >> 20) & 7;
    const bool signed_lanes = (variant & WHVI_FWHT_SIGNED_LANES) != 0;
    variant &= 0xFFFFF;
    const bool passes_only = long_form == LONG_PASSES || long_form == LONG_PASSES_UNGROUPED;
    const bool block_rows = log2d > max_single_pass_log2d<T>() && !(passes_only && sizeof(T) != 2);
    // (Rows of exactly two tiles -- f32 / i32 D = 8192, f64 D = 4096 -- stay with one 128-VGPR tile per wave: as 2-wave
    // blocks f32 gains 4 % (5.58 -> 5.80 TB/s), i32 loses 5 % (6.07 -> 5.74), f64 loses 1.5 %; profiles/r02/block_rows_w2.log.)
    if (block_rows && log2d <= max_block_log2d<T>()) {
        switch (log2d - multi_pass_low_log2d<T>()) {
        case 2: launch_block_rows<T, 2>(dst, src, rows, long_form, st); break;
        case 3: launch_block_rows<T, 3>(dst, src, rows, long_form, st); break;
        default: launch_block_rows<T, 4>(dst, src, rows, long_form, st); break;
        }
        return after_launch("fwht (block rows)");
    }
    if constexpr (sizeof(T) != 2) if (log2d > max_single_pass_log2d<T>()) {
        // Rows beyond one block: pass 1 = index bits [0, low) on contiguous pieces (src -> dst: 2^(LOW + 4)-element pieces,
        // a block each, unless the switch above asks for single tiles), passes 2.. = bits [low, log2d), up to HBMAX at a
        // time, in place on dst.  The passes run over row groups of <= 128 MiB one after the other, with cached
        // accesses, so every pass after the first finds its input in the 256 MiB Infinity Cache and HBM sees about one
        // read and one write of the data (LONG_PASSES_UNGROUPED: the whole buffer per pass, as in round 1).
        constexpr int LOW = multi_pass_low_log2d<T>();
        const int low = block_rows ? max_block_log2d<T>() : LOW;
        constexpr int HBMAX = 4;                               // 2^4 chunks = 64 accumulator VGPRs per thread
        constexpr int LV = ilog2(VEC);
        const int64_t group_bytes = long_form == LONG_PASSES_UNGROUPED ? 0 : (int64_t)128 << 20;
        const int64_t row_bytes = (int64_t)sizeof(T) << log2d;
        int64_t group_rows = group_bytes > 0 ? group_bytes / row_bytes : rows;
        if (group_rows < 1) group_rows = 1;
        for (int64_t r0 = 0; r0 < rows; r0 += group_rows) {
            const int64_t n = rows - r0 < group_rows ? rows - r0 : group_rows;
            char *gd = (char *)dst + r0 * row_bytes;
            const char *gs = (const char *)src + r0 * row_bytes;
            const int64_t g_chunks = (n << log2d) / VEC;
            if (block_rows) launch_block_rows<T, 4>(gd, gs, (n << log2d) >> low, long_form, st);
            else launch_rows<T, LOW, pick_k<T, LOW>(), false>(gd, gs, g_chunks, 0, st);
            for (int b0 = low; b0 < log2d;) {
                const int hb = (log2d - b0 < HBMAX) ? (log2d - b0) : HBMAX;
                const int64_t n_groups = g_chunks >> hb;
                const unsigned grid = (unsigned)((n_groups + 255) / 256);
                u32x4 *d = (u32x4 *)gd;
#define WHVI_HIGH(HB) hipLaunchKernelGGL((fwht_high_kernel<T, HB>), dim3(grid), dim3(256), 0, st, d, d, n_groups, b0 - LV)
                switch (hb) {
                case 1: WHVI_HIGH(1); break;
                case 2: WHVI_HIGH(2); break;
                case 3: WHVI_HIGH(3); break;
                default:
                    if constexpr (HBMAX >= 4) WHVI_HIGH(4);
                    break;
                }
#undef WHVI_HIGH
                b0 += hb;
            }
        }
        return after_launch("fwht (multi-pass)");
    }

#define WHVI_CASE(L)                                                                                 \
    case L:                                                                                          \
        if constexpr (L <= max_single_pass_log2d<T>()) {                                             \
            if (n_chunks > 0)                                                                        \
                launch_rows<T, L, pick_k<T, L>(), (TUNABLE && L >= 9 && L <= 12)>(dst, src, n_chunks, \
                                                                                  variant, st, signed_lanes); \
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
template <typename T, int LOG2D, int K = pick_k<T, LOG2D>()>
inline void launch_fused(void *dst, const void *src, const void *a, const void *b, const void *c,
                         int64_t rows, int64_t n_samples, int64_t sample_stride, int64_t group_rows,
                         int axis, int flags, hipStream_t st)
{
    constexpr int VEC = Elem<T>::VEC;
    constexpr bool SMALL_TILE = tile_vgprs<T, K>() <= 64;
    const int64_t n_chunks = (rows << LOG2D) / VEC;
    const int64_t n_tiles = (n_chunks + 64 * K - 1) / (64 * K);
    const FastDiv ds = make_fastdiv((uint32_t)sample_stride), dn = make_fastdiv((uint32_t)n_samples),
                  dg = make_fastdiv((uint32_t)group_rows);
    // "big": enough tiles to fill the chip several times over (32 per CU; 16 for the 32 KiB tiles of one f64 row of 4096)
    const bool big = n_tiles >= (int64_t)(SMALL_TILE ? 32 : 16) * num_cu();
    // (a shared source is small and cache-resident by intent: the streamed bytes are the destination's)
    const bool nt = big && stream_sized(n_chunks * 16, dst, (flags & WHVI_FUSED_SRC_SHARED) ? nullptr : src);
#define WHVI_FUSED(AX, EYE, NT, BLK, POL, STG)                                                          \
    do {                                                                                                \
        if constexpr (AX == WHVI_AXIS_COL && !(EYE) && POL == POLICY_DPP && LOG2D - ilog2(VEC) >= 6 && (STG) != STAGE_ABC) { \
            if (flags & (WHVI_FUSED_SRC_SHARED | WHVI_FUSED_ONE_TRANSFORM)) {                           \
                constexpr size_t smem_s = (((STG) == STAGE_AC ? 2 : 0) * sizeof(typename Elem<T>::acc) << LOG2D); \
                const bool shared_ = (flags & WHVI_FUSED_SRC_SHARED) != 0, one_ = (flags & WHVI_FUSED_ONE_TRANSFORM) != 0; \
                note_launch<T>("fused_shs_kernel", LOG2D, K, (int)AX, (bool)EYE, (bool)NT, (int)BLK, (int)POL, (int)STG, shared_, one_); \
                const dim3 grid_((unsigned)((n_tiles + (BLK / 64) - 1) / (BLK / 64)));                  \
                if (shared_ && one_)                                                                    \
                    hipLaunchKernelGGL((fused_shs_kernel<T, LOG2D, K, AX, EYE, NT, BLK, POL, STG, true, true>), grid_, dim3(BLK), smem_s, st, \
                                       (u32x4 *)dst, (const u32x4 *)src, (const T *)a, (const T *)b, (const T *)c, n_chunks, n_tiles, ds, dn, dg, flags, same_sample_blocks); \
                else if (shared_)                                                                       \
                    hipLaunchKernelGGL((fused_shs_kernel<T, LOG2D, K, AX, EYE, NT, BLK, POL, STG, true, false>), grid_, dim3(BLK), smem_s, st, \
                                       (u32x4 *)dst, (const u32x4 *)src, (const T *)a, (const T *)b, (const T *)c, n_chunks, n_tiles, ds, dn, dg, flags, same_sample_blocks); \
                else                                                                                    \
                    hipLaunchKernelGGL((fused_shs_kernel<T, LOG2D, K, AX, EYE, NT, BLK, POL, STG, false, true>), grid_, dim3(BLK), smem_s, st, \
                                       (u32x4 *)dst, (const u32x4 *)src, (const T *)a, (const T *)b, (const T *)c, n_chunks, n_tiles, ds, dn, dg, flags, same_sample_blocks); \
                break;                                                                                  \
            }                                                                                           \
        }                                                                                               \
        note_launch<T>("fused_shs_kernel", LOG2D, K, (int)AX, (bool)EYE, (bool)NT, (int)BLK, (int)POL, (int)STG, false, false); \
        constexpr size_t smem = ((POL == POLICY_LDS) ? (size_t)(BLK / 64) * slab_bytes : 0) +           \
                                (((STG) == STAGE_ABC ? 3 : ((STG) == STAGE_AC ? 2 : 0)) * sizeof(typename Elem<T>::acc) << LOG2D); \
        if constexpr (smem > 64 * 1024)   /* per launch: the attribute belongs to the CURRENT device's copy */ \
            (void)hipFuncSetAttribute((const void *)fused_shs_kernel<T, LOG2D, K, AX, EYE, NT, BLK, POL, STG>, \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);           \
        hipLaunchKernelGGL((fused_shs_kernel<T, LOG2D, K, AX, EYE, NT, BLK, POL, STG>),                 \
                           dim3((unsigned)((n_tiles + (BLK / 64) - 1) / (BLK / 64))), dim3(BLK), smem, st, \
                           (u32x4 *)dst, (const u32x4 *)src, (const T *)a, (const T *)b, (const T *)c,  \
                           n_chunks, n_tiles, ds, dn, dg, flags, same_sample_blocks);                   \
    } while (0)
    uint32_t same_sample_blocks = 0;
    constexpr bool LDS_OK = sizeof(typename Elem<T>::acc) == 4 && K * VEC == 64;
    constexpr size_t slab_bytes = (size_t)K * (64 * VEC + VEC) * 4;   // lds_slab_floats<VEC, K>() * 4
    (void)LDS_OK;
    // Column-axis launch of big problems, measured on MI355X (round 1, tools/tune_fused.py, D = 2048 / 4096,
    // 4 GiB): 256-thread blocks beat 512 (12 vs 8 waves per CU at ~200 VGPRs), staging the shared a / c
    // vectors in LDS is worth +5 %, non-temporal data accesses +4 %, and the LDS-staged butterfly
    // network ties the DPP one (the kernel is bound by its 4x load-instruction stream, not by VALU):
    //   dpp/256/nt/staged 5.05 TB/s | lds/256 5.00 | dpp/256 4.70 | dpp/512/nt 4.04 | lds/512/nt 4.40
    //   (one 8 KiB row per wave at D = 2048 -- 106 VGPRs, twice the waves -- ties at 4.87; 1024-thread blocks 4.4)
    // Round 3 -- which vectors a block stages in LDS (kernels.hpp: STAGE_*):
    //   * rows in (sample, batch, D) order with sample_stride a multiple of the rows per block: every block lies inside
    //     ONE sample, so that sample's a and c are staged (STAGE_AC) also when they are per-sample: 6.35-6.40 TB/s for f32
    //     at D = 512 .. 4096 against 5.4-5.7 with the three vectors from L2;
    //   * otherwise shared a / c are staged and per-sample ones come from L2 (STAGE_NONE).
    //   Staging b as well (STAGE_ABC, tuning builds) ties at D = 4096 f32 / D = 2048 f64 (6.41 vs 6.38, 6.38 vs 6.34) and
    //   LOSES at D <= 2048 f32 (5.76 vs 6.39) and for f64 rows of 4096 (96 KiB of LDS: one block per CU, 4.1 vs 5.8).
    // Tuning builds only: WHVI_FUSED_TUNE=<policy 0|2><block 2|5 (unused: always 256)><nt 0|1><stage 0|1|3> overrides (read once).
    constexpr int64_t rows_per_block = ((int64_t)4 * 64 * K * VEC) >> LOG2D;       // >= 1 for every staged shape (LOG2D <= 12)
    const bool one_sample_blocks = rows_per_block >= 1 && sample_stride % rows_per_block == 0;
#ifdef WHVI_TUNING_BUILD
    static const char *tune_env = [] {          // ignored unless it is exactly four digits: never read past the NUL
        const char *e = WHVI_TUNE_ENV("WHVI_FUSED_TUNE");
        if (e == nullptr || strlen(e) != 4) return (const char *)nullptr;
        for (int i = 0; i < 4; ++i)
            if (e[i] < '0' || e[i] > '9') return (const char *)nullptr;
        return e;
    }();
#else
    constexpr const char *tune_env = nullptr;
#endif
    int stage = STAGE_NONE;
    if ((flags & WHVI_FUSED_SRC_SHARED) && rows_per_block >= 1 && sample_stride % rows_per_block == 0 &&
        rows == n_samples * sample_stride && ((sample_stride / rows_per_block) & 7) == 0 && n_samples > 1)
        same_sample_blocks = 2;        // shared source: sample index fastest within an XCD (kernels.hpp)
    if (big) {
        if (one_sample_blocks || (flags & (WHVI_FUSED_A_PER_SAMPLE | WHVI_FUSED_C_PER_SAMPLE)) == 0) stage = STAGE_AC;
        // per-sample a / c with rows in (batch, sample, D) order and one 128-register row per tile (f64 D = 4096): let every
        // block take four rows of the same sample, S rows apart, and stage that sample's vectors (kernels.hpp).  Measured
        // (4 GiB, 64 samples): 3.83 -> 4.63 TB/s.  NOT for 64-register one-row tiles (f32 D = 4096, f64 D = 2048): there
        // the four rows of a block being 1 MiB apart instead of adjacent costs what the staging gains (5.42 vs 5.55 and
        // 5.42 vs 5.42 TB/s; the same kernel on contiguous blocks with shared vectors runs at 6.31)
        else if (!SMALL_TILE && rows_per_block == 4 && sample_stride == 1 && n_samples > 1 && axis == WHVI_AXIS_COL &&
                 src != nullptr && rows % (4 * n_samples) == 0 && n_tiles == rows) {
            stage = STAGE_AC;
            same_sample_blocks = 1;
        }
    }
    bool use_nt = nt;
    if (tune_env != nullptr) {
        use_nt = tune_env[2] != '0';
        const int want = tune_env[3] - '0';
        if (want == STAGE_NONE || (want == STAGE_AC && ((flags & 3) == 0 || one_sample_blocks)) || (want == STAGE_ABC && one_sample_blocks && !(flags & (WHVI_FUSED_SRC_SHARED | WHVI_FUSED_ONE_TRANSFORM)))) {
            stage = want;
            same_sample_blocks = 0;
        }
    }
#ifdef WHVI_TUNING_BUILD
#define WHVI_FUSED_ABC(AX, EYE, POL)                                                                    \
    if (stage == STAGE_ABC) { if (use_nt) WHVI_FUSED(AX, EYE, true, 256, POL, STAGE_ABC); else WHVI_FUSED(AX, EYE, false, 256, POL, STAGE_ABC); break; }
#else
#define WHVI_FUSED_ABC(AX, EYE, POL)
#endif
#define WHVI_FUSED_STAGED(AX, EYE, POL)                                                                 \
    do {                                                                                                \
        WHVI_FUSED_ABC(AX, EYE, POL);                                                                   \
        if (stage == STAGE_AC) { if (use_nt) WHVI_FUSED(AX, EYE, true, 256, POL, STAGE_AC); else WHVI_FUSED(AX, EYE, false, 256, POL, STAGE_AC); } \
        else if (use_nt) WHVI_FUSED(AX, EYE, true, 256, POL, STAGE_NONE);                               \
        else WHVI_FUSED(AX, EYE, false, 256, POL, STAGE_NONE);                                          \
    } while (0)
#define WHVI_FUSED_GEOM(AX, EYE)                                                                        \
    do {                                                                                                \
        if constexpr (AX == WHVI_AXIS_COL && !(EYE) && sizeof(T) >= 4 && LOG2D >= 9 && LOG2D <= 12) {   \
            if (big) {                                                                                  \
                WHVI_FUSED_TUNED_POLICY(AX, EYE);                                                       \
                WHVI_FUSED_STAGED(AX, EYE, POLICY_DPP);                                                 \
                break;                                                                                  \
            }                                                                                           \
        }                                                                                               \
        if (nt) WHVI_FUSED(AX, EYE, true, 256, POLICY_DPP, STAGE_NONE);                                 \
        else WHVI_FUSED(AX, EYE, false, 256, POLICY_DPP, STAGE_NONE);                                   \
    } while (0)
#ifdef WHVI_TUNING_BUILD     /* the LDS-staged butterfly network as an A/B (f32, 64-register tiles) */
#define WHVI_FUSED_TUNED_POLICY(AX, EYE)                                                                \
    if constexpr (LDS_OK) {                                                                             \
        if (tune_env != nullptr && tune_env[0] == '2') { WHVI_FUSED_STAGED(AX, EYE, POLICY_LDS); break; } \
    }
#else
#define WHVI_FUSED_TUNED_POLICY(AX, EYE)
#endif
    if (src == nullptr) WHVI_FUSED_GEOM(WHVI_AXIS_ROW, true);
    else if (axis == WHVI_AXIS_ROW) WHVI_FUSED_GEOM(WHVI_AXIS_ROW, false);
    else WHVI_FUSED_GEOM(WHVI_AXIS_COL, false);
#undef WHVI_FUSED_GEOM
#undef WHVI_FUSED_TUNED_POLICY
#undef WHVI_FUSED_STAGED
#undef WHVI_FUSED_ABC
#undef WHVI_FUSED
}

template <typename T>
inline int fused_dispatch(void *dst, const void *src, const void *a, const void *b, const void *c,
                          int64_t rows, int32_t log2d, int64_t n_samples, int64_t sample_stride,
                          int64_t group_rows, int32_t axis, int32_t flags, void *stream)
{
    constexpr int LV = ilog2(Elem<T>::VEC);
    // a shared source (WHVI_FUSED_SRC_SHARED) is sample_stride rows long, not `rows`: its overlap check is done below
    const bool shared_src = (flags & WHVI_FUSED_SRC_SHARED) != 0 && src != nullptr;
    int rc = check_common(dst, shared_src ? nullptr : src, rows, log2d, max_single_pass_log2d<T>(), sizeof(T), true);
    if (rc != WHVI_OK) return rc;
    if (shared_src && rows > 0) {
        if ((uintptr_t)src & 15) return fail(WHVI_ERR_ALIGN, "whvi: %s pointer is not 16-byte aligned", "src");
        const char *d = (const char *)dst, *sp = (const char *)src;
        const int64_t dbytes = (rows << log2d) * (int64_t)sizeof(T), sbytes = (sample_stride << log2d) * (int64_t)sizeof(T);
        if (sample_stride >= 1 && d < sp + sbytes && sp < d + dbytes)
            return fail(WHVI_ERR_OVERLAP, "whvi: dst overlaps the shared source%s", "");
    }
    if (axis != WHVI_AXIS_ROW && axis != WHVI_AXIS_COL)
        return fail(WHVI_ERR_ARG, "whvi: bad axis%s %lld", "", axis);
    if (flags & ~(WHVI_FUSED_A_PER_SAMPLE | WHVI_FUSED_C_PER_SAMPLE | WHVI_FUSED_SRC_SHARED | WHVI_FUSED_ONE_TRANSFORM))
        return fail(WHVI_ERR_ARG, "whvi: unknown fused flags%s 0x%llx", "", flags);
    if ((flags & WHVI_FUSED_ONE_TRANSFORM) && (src == nullptr || c != nullptr || axis != WHVI_AXIS_COL || log2d - LV < 6))
        return fail(WHVI_ERR_ARG, "whvi: the one-transform flag needs axis = COL, src != NULL, c == NULL and rows of at least "
                    "64 sixteen-byte chunks%s", "");
    if ((flags & WHVI_FUSED_SRC_SHARED) && (src == nullptr || dst == src || axis != WHVI_AXIS_COL || log2d - LV < 6))
        return fail(WHVI_ERR_ARG, "whvi: the shared-source flag needs axis = COL, src != NULL, dst != src and rows of at least "
                    "64 sixteen-byte chunks%s", "");
    if (n_samples < 1 || sample_stride < 1 || group_rows < 1)
        return fail(WHVI_ERR_ARG, "whvi: n_samples, sample_stride and group_rows must be >= 1%s", "");
    if (rows >= ((int64_t)1 << 32) || n_samples >= ((int64_t)1 << 32) || sample_stride >= ((int64_t)1 << 32) ||
        group_rows >= ((int64_t)1 << 32))
        return fail(WHVI_ERR_SIZE, "whvi: the fused pipeline indexes rows with 32 bits%s", "");
    if (src == nullptr && (axis != WHVI_AXIS_ROW || group_rows > ((int64_t)1 << log2d)))
        return fail(WHVI_ERR_ARG, "whvi: src == NULL (identity input) needs axis = ROW and group_rows <= D%s", "");
    if (axis == WHVI_AXIS_COL && (((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)c & 15)))
        return fail(WHVI_ERR_ALIGN, "whvi: column scale vectors must be 16-byte aligned%s", "");
    if (rows == 0) return WHVI_OK;
    hipStream_t st = (hipStream_t)stream;
    if (log2d < LV) {   // rows shorter than one chunk: thread-per-row kernel
        const FastDiv ds = make_fastdiv((uint32_t)sample_stride), dn = make_fastdiv((uint32_t)n_samples),
                      dg = make_fastdiv((uint32_t)group_rows);
        const unsigned grid = (unsigned)((rows + 255) / 256);
#define WHVI_SMALL(L, AX, EYE)                                                                             \
    hipLaunchKernelGGL((fused_small_kernel<T, L, AX, EYE>), dim3(grid), dim3(256), 0, st, (T *)dst,        \
                       (const T *)src, (const T *)a, (const T *)b, (const T *)c, rows, ds, dn, dg, flags)
#define WHVI_SMALL_L(L)                                                    \
    do {                                                                   \
        if (src == nullptr) WHVI_SMALL(L, WHVI_AXIS_ROW, true);            \
        else if (axis == WHVI_AXIS_ROW) WHVI_SMALL(L, WHVI_AXIS_ROW, false); \
        else WHVI_SMALL(L, WHVI_AXIS_COL, false);                          \
    } while (0)
        if (log2d == 0) WHVI_SMALL_L(0);
        else if constexpr (LV >= 2) { if (log2d == 1) WHVI_SMALL_L(1); }
#undef WHVI_SMALL_L
#undef WHVI_SMALL
        return after_launch("fused_shs (short rows)");
    }
#define WHVI_CASE(L)                                                                                     \
    case L:                                                                                              \
        if constexpr (L >= LV && L <= max_single_pass_log2d<T>())                                        \
            launch_fused<T, L>(dst, src, a, b, c, rows, n_samples, sample_stride, group_rows, axis, flags, st); \
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
