// whvi_amd/csrc/fwht_tile.hpp -- register-resident Walsh-Hadamard butterfly network for one
// wavefront-owned tile on gfx950 (CDNA4, 64-lane waves).
//
// Computes what the reference's fwht_batch1_kernel computes (src/fwht/cuda/fwht_cuda_kernel.cu:
// 74-146) but in the stage order of the reference C++ oracle (src/fwht/cpp/fwht.cpp:7-18,
// strides 1, 2, 4, ...), so f32/f64 results are bit-identical to that oracle.
//
// Data layout (chosen for coalesced 16-byte-per-lane HBM access, no LDS):
//   a tile is 64 * K chunks of 16 bytes; lane l owns chunks k*64 + l, k = 0..K-1, i.e. one
//   wave-instruction global_load_dwordx4 reads 1 KiB contiguous.  With VEC elements per
//   chunk the element index inside the tile is
//         idx = k * (64*VEC) + l * VEC + c          c in [0,VEC), l in [0,64), k in [0,K)
//   so index bits [0,LV) live in registers (c), bits [LV,LV+6) are the LANE id, bits
//   [LV+6, LV+6+LK) live in registers again (k).  Rows are 2^LOG2D <= tile elements, never
//   split across tiles; index bits >= LOG2D select the row and are never butterflied.
//
// Stage s (index bit s), always in ascending s:
//   s <  LV        : in-register add/sub on (c, c ^ 2^s)
//   lane bit 0,1   : DPP quad_perm        (v_add_f32_dpp with the sign folded in by v_xor)
//   lane bit 2     : DPP row_half_mirror o quad_perm[3,2,1,0]  (= lane ^ 4)
//   lane bit 3     : DPP row_ror:8        (= lane ^ 8)
//   lane bit 4,5   : v_permlane16_swap / v_permlane32_swap turn the lane bit into a REGISTER
//                    bit (a 2x2 transpose between a k-pair of registers and the lane bit), the
//                    butterfly is then a plain in-register add/sub.  The layout stays
//                    transposed until the partner k-bit's own stage, which repeats the same
//                    swap (and thereby restores the layout) -- or is swapped back at once when
//                    that k-bit is a row bit.
//   s >= LV+6      : in-register add/sub on (k, k ^ 2^j)
// POLICY_SHFL replaces every cross-lane step by ds_bpermute (__shfl_xor); it exists to
// cross-check the DPP/permlane paths on hardware.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "tuning.hpp"

namespace whvi {

constexpr int POLICY_DPP = 0;
constexpr int POLICY_SHFL = 1;
constexpr int POLICY_LDS = 2;    // lane-bit stages through one LDS transpose (see fwht_tile_lds)

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

template <int I> using IC = std::integral_constant<int, I>;

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

// ---- 32-bit cross-lane moves -------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}

// value of lane (self ^ (1 << LB)), LB in [0,3], through the DPP network only
template <int LB>
__device__ __forceinline__ uint32_t dpp_xor_u32(uint32_t v)
{
    static_assert(LB >= 0 && LB <= 3, "DPP covers lane bits 0..3");
    if constexpr (LB == 0) return dpp_u32<0xB1>(v);               // quad_perm:[1,0,3,2]
    else if constexpr (LB == 1) return dpp_u32<0x4E>(v);          // quad_perm:[2,3,0,1]
    else if constexpr (LB == 2) return dpp_u32<0x1B>(dpp_u32<0x141>(v)); // half_mirror (l^7) then [3,2,1,0] (l^3)
    else return dpp_u32<0x128>(v);                                // row_ror:8
}

template <typename A> struct Bits;
template <> struct Bits<float> {
    template <int LB> static __device__ __forceinline__ float partner_dpp(float v)
    {
        float p = __builtin_bit_cast(float, dpp_xor_u32<LB>(__builtin_bit_cast(uint32_t, v)));
#ifdef WHVI_EXP_UNFUSED_DPP          // tuning builds: keep v_mov_b32_dpp apart from the add that consumes it (the i32 code shape)
        asm volatile("" : "+v"(p));
#endif
        return p;
    }
    // upper lane of the pair computes partner - v, lower lane v + partner; folding the sign
    // into v first keeps it at one v_xor + one (DPP-fused) v_add.  p + (-v) == p - v exactly.
    static __device__ __forceinline__ float combine(float v, float p, uint32_t sign_mask, bool)
    {
        return p + __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) ^ sign_mask);
    }
    static __device__ __forceinline__ float fold_sign(float v, uint32_t sign_mask)
    {
        return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) ^ sign_mask);
    }
    static __device__ __forceinline__ float add_folded(float p, float, float folded, bool) { return p + folded; }
};
template <> struct Bits<int32_t> {
    template <int LB> static __device__ __forceinline__ int32_t partner_dpp(int32_t v)
    {
        return (int32_t)dpp_xor_u32<LB>((uint32_t)v);
    }
    static __device__ __forceinline__ int32_t combine(int32_t v, int32_t p, uint32_t, bool upper)
    {
        // wraps on overflow exactly like the reference's int tensors (unsigned arithmetic)
        uint32_t uv = (uint32_t)v, up = (uint32_t)p;
        return (int32_t)(upper ? up - uv : uv + up);
    }
    // two's complement: -v = (v ^ m) - m with m = all ones on the upper lane
    static __device__ __forceinline__ int32_t fold_sign(int32_t v, uint32_t sign_mask)
    {
        const uint32_t m = (uint32_t)((int32_t)sign_mask >> 31);
        return (int32_t)((((uint32_t)v) ^ m) - m);
    }
    static __device__ __forceinline__ int32_t add_folded(int32_t p, int32_t, int32_t folded, bool)
    {
        return (int32_t)((uint32_t)p + (uint32_t)folded);
    }
};
template <> struct Bits<double> {
    template <int LB> static __device__ __forceinline__ double partner_dpp(double v)
    {
        uint64_t u = __builtin_bit_cast(uint64_t, v);
        uint32_t lo = dpp_xor_u32<LB>((uint32_t)u);
        uint32_t hi = dpp_xor_u32<LB>((uint32_t)(u >> 32));
        return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    }
    static __device__ __forceinline__ double combine(double v, double p, uint32_t sign_mask, bool)
    {
        uint64_t u = __builtin_bit_cast(uint64_t, v) ^ ((uint64_t)sign_mask << 32);
        return p + __builtin_bit_cast(double, u);
    }
    static __device__ __forceinline__ double fold_sign(double v, uint32_t sign_mask)
    {
        return __builtin_bit_cast(double, __builtin_bit_cast(uint64_t, v) ^ ((uint64_t)sign_mask << 32));
    }
    static __device__ __forceinline__ double add_folded(double p, double, double folded, bool) { return p + folded; }
};

// v_permlane{16,32}_swap on one 32-bit register pair: afterwards
//   a' = [a.even_part, b.even_part],  b' = [a.odd_part, b.odd_part]
// where "part" is a 16-lane row (W=16) or a 32-lane half (W=32).
template <int W>
__device__ __forceinline__ void swap_u32(uint32_t &a, uint32_t &b)
{
    if constexpr (W == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r[0];
        b = r[1];
    } else {
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        a = r[0];
        b = r[1];
    }
}

template <int W, typename A>
__device__ __forceinline__ void swap_pair(A &a, A &b)
{
    if constexpr (sizeof(A) == 4) {
        uint32_t ua = __builtin_bit_cast(uint32_t, a), ub = __builtin_bit_cast(uint32_t, b);
        swap_u32<W>(ua, ub);
        a = __builtin_bit_cast(A, ua);
        b = __builtin_bit_cast(A, ub);
    } else {
        uint64_t ua = __builtin_bit_cast(uint64_t, a), ub = __builtin_bit_cast(uint64_t, b);
        uint32_t al = (uint32_t)ua, ah = (uint32_t)(ua >> 32), bl = (uint32_t)ub, bh = (uint32_t)(ub >> 32);
        swap_u32<W>(al, bl);
        swap_u32<W>(ah, bh);
        a = __builtin_bit_cast(A, ((uint64_t)ah << 32) | al);
        b = __builtin_bit_cast(A, ((uint64_t)bh << 32) | bl);
    }
}

// radix-2 butterfly exactly as src/fwht/cpp/fwht.cpp:11-13: (lo, hi) -> (lo + hi, lo - hi)
template <typename A>
__device__ __forceinline__ void bfly(A &lo, A &hi)
{
    if constexpr (std::is_same<A, int32_t>::value) {
        uint32_t a = (uint32_t)lo, b = (uint32_t)hi;
        lo = (int32_t)(a + b);
        hi = (int32_t)(a - b);
    } else {
        A a = lo, b = hi;
        lo = a + b;
        hi = a - b;
    }
}

// Two independent butterflies on adjacent registers as ONE v_pk_add_f32 pair (f32 only): halves
// the issue slots of every in-register stage.  Same IEEE adds/subs, so the bits do not change.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// -DWHVI_NO_PK (tuning builds): no explicit packed f32 instructions anywhere; combine with -fno-slp-vectorize so the
// compiler forms none of its own either
#ifdef WHVI_NO_PK
constexpr bool kPackedF32 = false;
#else
constexpr bool kPackedF32 = true;
#endif

template <typename A, bool PK = true>
__device__ __forceinline__ void bfly2(A &lo0, A &lo1, A &hi0, A &hi1)
{
    if constexpr (std::is_same<A, float>::value && PK && kPackedF32) {
        f32x2 a = {lo0, lo1}, b = {hi0, hi1};
        f32x2 s = a + b, d = a - b;
        lo0 = s[0];
        lo1 = s[1];
        hi0 = d[0];
        hi1 = d[1];
    } else {
        bfly(lo0, hi0);
        bfly(lo1, hi1);
    }
}

// Two independent products / sums as ONE v_pk_mul_f32 / v_pk_add_f32 (f32; plain ops otherwise): (x0, x1) *= (m0, m1)
// and (a0, a1) += (x0, x1).  Separate IEEE roundings per element, like the scalar forms.
template <typename A>
__device__ __forceinline__ void mul2(A &x0, A &x1, A m0, A m1)
{
    if constexpr (std::is_same<A, float>::value && kPackedF32) {
        f32x2 x = {x0, x1}, m = {m0, m1};
        x = m * x;
        x0 = x[0];
        x1 = x[1];
    } else {
        x0 = m0 * x0;
        x1 = m1 * x1;
    }
}
template <typename A>
__device__ __forceinline__ void add2(A &a0, A &a1, A x0, A x1)
{
    if constexpr (std::is_same<A, float>::value && kPackedF32) {
        f32x2 a = {a0, a1}, x = {x0, x1};
        a = a + x;
        a0 = a[0];
        a1 = a[1];
    } else {
        a0 = a0 + x0;
        a1 = a1 + x1;
    }
}

// One full FWHT of every 2^LOG2D-element row held in r[K][VEC] (layout above).
// PK (bit mask: 1 = in-chunk stages, 2 = permlane-swap stages, 4 = k-bit stages): issue those stages of f32 tiles as v_pk_add_f32 pairs.  Same bits either way.  Packed adds halve
// the issue slots of those stages but want even-aligned register pairs: in the plain streaming kernel that costs
// 38 VGPRs (145 vs 107 = 3 vs 4 waves per SIMD) and 1.5 % of the stream at D = 4096, so it passes PK = false; the
// fused / weight kernels (more VALU work per byte) are faster with it.
//
// SIGNED (f32 / f64, POLICY_DPP only): the DPP lane stages as ONE fused multiply-add per element instead of a sign fold
// plus an add.  Every lane computes  own + s * partner  with s = +/-1 (exact: the same bits as own +/- partner), which is
// the butterfly's result in the lower lane of a pair and its NEGATIVE in the upper lane; instead of repairing that, the
// tile is allowed to hold sigma(lane) * value with sigma = (-1)^popcount(lane & mask): a stage on lane bit b uses
// s = +1 / -1 for lower / upper lanes when the incoming mask has bit b clear (-1 / +1 when set) and toggles bit b of the
// mask.  Everything else in the network is sign-agnostic per lane (in-register butterflies: (-a) +/- (-b) = -(a +/- b)
// exactly; permlane swaps pair lanes with equal low four bits, i.e. equal sigma; elementwise scalings commute with it).
// SIGN_IN is the mask the tile arrives with; fwht_sign_out() the one it leaves with.  Two transforms in a row (the fused
// pipeline) toggle the same bits twice and end with mask 0: no repair at all.  Exact cancellations give +0 in either
// convention, so results stay bit-identical for inputs without negative zeros.
template <int VEC, int LOG2D>
constexpr int fwht_sign_out(int sign_in)
{
    constexpr int LV = ilog2(VEC);
    constexpr int NL = (LOG2D - LV) < 0 ? 0 : ((LOG2D - LV) > 4 ? 4 : (LOG2D - LV));      // DPP lane stages of this row length
    return sign_in ^ ((1 << NL) - 1);
}
// x[i] += s * x[i](lane ^ (1 << LB)) for eight registers: v_fmac_f32_dpp reads the partner lane's value of its own
// destination register.  A DPP instruction must not read a VGPR within two wait states of a VALU write to it and the
// compiler's hazard recogniser cannot see into the block, so every block opens with s_nop 1 (covers whatever wrote the
// eight registers just before) and touches eight DIFFERENT registers; lane ^ 4 needs two hops (row_half_mirror, then
// quad_perm [3,2,1,0]) through temporaries written eight instructions before they are read.
#define WHVI_FMAC_DPP8(CTRL)                                                                                   \
    asm volatile("s_nop 1\n\t"                                                                                 \
                 "v_fmac_f32_dpp %0, %0, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                            \
                 "v_fmac_f32_dpp %1, %1, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                            \
                 "v_fmac_f32_dpp %2, %2, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                            \
                 "v_fmac_f32_dpp %3, %3, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                            \
                 "v_fmac_f32_dpp %4, %4, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                            \
                 "v_fmac_f32_dpp %5, %5, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                            \
                 "v_fmac_f32_dpp %6, %6, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                            \
                 "v_fmac_f32_dpp %7, %7, %8 " CTRL " row_mask:0xf bank_mask:0xf"                                 \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
                 : "v"(s))
template <int LB>
__device__ __forceinline__ void fmac_dpp8(float *x, float s)
{
    static_assert(LB >= 0 && LB <= 3, "DPP covers lane bits 0..3");
    if constexpr (LB == 0) WHVI_FMAC_DPP8("quad_perm:[1,0,3,2]");
    else if constexpr (LB == 1) WHVI_FMAC_DPP8("quad_perm:[2,3,0,1]");
    else if constexpr (LB == 3) WHVI_FMAC_DPP8("row_ror:8");
    else {
        float t0, t1, t2, t3, t4, t5, t6, t7;
        asm volatile("s_nop 1\n\t"
                     "v_mov_b32_dpp %8, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %9, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %10, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %11, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %12, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %13, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %14, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %15, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %0, %8, %16 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %1, %9, %16 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %2, %10, %16 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %3, %11, %16 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %4, %12, %16 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %5, %13, %16 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %6, %14, %16 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %7, %15, %16 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                       "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                     : "v"(s));
    }
}
#undef WHVI_FMAC_DPP8
__device__ __forceinline__ float fma_pm(float p, float s, float v) { return __builtin_fmaf(p, s, v); }
__device__ __forceinline__ double fma_pm(double p, double s, double v) { return __builtin_fma(p, s, v); }

template <typename A, int VEC, int K, int LOG2D, int POLICY, int PK = 7, bool SIGNED = false, int SIGN_IN = 0>
__device__ __forceinline__ void fwht_tile(A (&r)[K][VEC], const int lane)
{
    static_assert(!SIGNED || (POLICY == POLICY_DPP && !std::is_same<A, int32_t>::value), "signed form: f32 / f64 DPP network");
    constexpr int LV = ilog2(VEC);
    constexpr int LK = ilog2(K);
    static_assert(LOG2D <= LV + 6 + LK, "row does not fit the tile");
    // partner k-bits of the permlane-swap stages and whether their own stage exists
    constexpr int KB5 = (K >= 4) ? 1 : 0;
    constexpr bool PAIR4 = (POLICY == POLICY_DPP) && (K >= 2) && (LV + 6 + 0 < LOG2D);
    constexpr bool PAIR5 = (POLICY == POLICY_DPP) && (K >= 4) && (LV + 6 + 1 < LOG2D);

    static_for<0, LOG2D>([&](auto s_) {
        constexpr int S = decltype(s_)::value;
        if constexpr (S < LV) {
            // ---- in-chunk register stage ----
            constexpr int H = 1 << S;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (H >= 2) {
#pragma unroll
                    for (int c = 0; c < VEC; c += 2)
                        if ((c & H) == 0) bfly2<A, (PK & 1) != 0>(r[k][c], r[k][c + 1], r[k][c | H], r[k][(c | H) + 1]);
                } else {
#pragma unroll
                    for (int c = 0; c < VEC; ++c)
                        if ((c & H) == 0) bfly(r[k][c], r[k][c | H]);
                }
            }
        } else if constexpr (S < LV + 6) {
            // ---- lane stage ----
            constexpr int LB = S - LV;
            const bool upper = (lane >> LB) & 1;
            const uint32_t sign_mask = upper ? 0x80000000u : 0u;
            if constexpr (POLICY == POLICY_SHFL || (LB >= 4 && K < 2)) {
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int c = 0; c < VEC; ++c) {
                        A p = __shfl_xor(r[k][c], 1 << LB, 64);
                        r[k][c] = Bits<A>::combine(r[k][c], p, sign_mask, upper);
                    }
            } else if constexpr (LB < 4 && SIGNED) {
                // own + s * partner, s = -1 where this lane's bit LB differs from the incoming sign mask's bit LB ... see above
                const A sgn = (upper != (bool)((SIGN_IN >> LB) & 1)) ? (A)-1 : (A)1;
                if constexpr (std::is_same<A, float>::value && (K * VEC) % 8 == 0) {
                    // v_fmac_f32_dpp, eight elements per block (the compiler fuses a DPP move into v_add_f32 but not into
                    // an fma: VOP3 has no DPP form on gfx9 and the fmac shrink comes after its DPP combiner)
                    float *flat = &r[0][0];
#pragma unroll
                    for (int g = 0; g < K * VEC; g += 8) fmac_dpp8<LB>(flat + g, sgn);
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k)
#pragma unroll
                        for (int c = 0; c < VEC; ++c)
                            r[k][c] = fma_pm(Bits<A>::template partner_dpp<LB>(r[k][c]), sgn, r[k][c]);
                }
            } else if constexpr (LB < 4) {
                // A DPP instruction needs two wait states after a VALU write of ANY of its VGPR
                // operands, so "v_xor t, mask, v; v_add_f32_dpp v, v, t" back to back costs an
                // s_nop each time.  Sign-fold a group of 8 elements first, then issue their 8
                // DPP adds: the distance covers the hazard and the nops disappear (-13 % issue
                // slots per tile).  sched_barrier pins the grouping against the scheduler.
                constexpr int G = (8 / VEC) > 0 ? (8 / VEC) : 1;   // chunks per group
#pragma unroll
                for (int k0 = 0; k0 < K; k0 += G) {
                    A t[G][VEC];
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int c = 0; c < VEC; ++c)
                            if (k0 + g < K) t[g][c] = Bits<A>::fold_sign(r[k0 + g][c], sign_mask);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int c = 0; c < VEC; ++c)
                            if (k0 + g < K)
                            {
                                A partner = Bits<A>::template partner_dpp<LB>(r[k0 + g][c]);
                                r[k0 + g][c] = Bits<A>::add_folded(partner, r[k0 + g][c], t[g][c], upper);
                            }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                constexpr int W = (LB == 4) ? 16 : 32;
                constexpr int KB = (LB == 4) ? 0 : KB5;
                constexpr bool PAIRED = (LB == 4) ? PAIR4 : PAIR5;
                constexpr int KH = 1 << KB;
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if ((k & KH) == 0) {
#pragma unroll
                        for (int c = 0; c < VEC; c += 2) {
                            swap_pair<W>(r[k][c], r[k | KH][c]);
                            swap_pair<W>(r[k][c + 1], r[k | KH][c + 1]);
                            bfly2<A, (PK & 2) != 0>(r[k][c], r[k][c + 1], r[k | KH][c], r[k | KH][c + 1]);
                            if constexpr (!PAIRED) {
                                swap_pair<W>(r[k][c], r[k | KH][c]);
                                swap_pair<W>(r[k][c + 1], r[k | KH][c + 1]);
                            }
                        }
                    }
            }
        } else {
            // ---- k-bit register stage ----
            constexpr int J = S - LV - 6;
            constexpr int KH = 1 << J;
            constexpr bool VIA16 = (J == 0) && PAIR4;    // this index bit currently sits on lane bit 4
            constexpr bool VIA32 = (J == KB5) && PAIR5;  // ... on lane bit 5
#pragma unroll
            for (int k = 0; k < K; ++k)
                if ((k & KH) == 0) {
#pragma unroll
                    for (int c = 0; c < VEC; c += 2) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if constexpr (VIA16) swap_pair<16>(r[k][c + h], r[k | KH][c + h]);
                            else if constexpr (VIA32) swap_pair<32>(r[k][c + h], r[k | KH][c + h]);
                        }
                        bfly2<A, (PK & 4) != 0>(r[k][c], r[k][c + 1], r[k | KH][c], r[k | KH][c + 1]);
                    }
                }
        }
    });
}

// ---- LDS-staged variant ----------------------------------------------------------------------
// The six lane-bit stages cost 2-3x an in-register stage on the VALU (sign fold + DPP add, or a
// permlane swap per pair).  Here the wave transposes its tile through a private LDS slab instead:
//
//   layout A (as loaded):  lane l, regs (k, c):      idx = k*64*VEC + l*VEC + c
//   layout B (transposed): lane l' = k'*VEC + c', regs j = 0..63:   idx = k'*64*VEC + j*VEC + c'
//
// so index bits [LV, LV+6) -- the lane id in layout A -- are REGISTER bits in layout B and all of
// their stages are plain packed adds.  A -> B: 16-byte ds_write of each chunk (linear rows, padded
// by one chunk), 64 ds_read_b32 at stride VEC; B -> A: the same in reverse.  Both directions are
// conflict-free: a row pitch of 64*VEC + VEC floats puts half-wave lane l' on bank (l' + j*VEC) % 32.
// Same adds in the same (ascending) order as the DPP network, so the bits are identical.
// The slab is private to the wave: no block barrier, only a wavefront-scope fence so the compiler
// keeps the write / read phases in order (the LDS processes one wave's DS operations in order).
template <int VEC, int K> constexpr int lds_slab_floats() { return K * (64 * VEC + VEC); }

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename A, int VEC, int K, int LOG2D>
__device__ __forceinline__ void fwht_tile_lds(A (&r)[K][VEC], const int lane, A *slab)
{
    static_assert(sizeof(A) == 4, "LDS-staged tile: 32-bit arithmetic types");
    static_assert(K * VEC == 64, "layout B gives every lane one (k, c) pair");
    constexpr int LV = ilog2(VEC);
    constexpr int PITCH = 64 * VEC + VEC;
    typedef A vec4 __attribute__((ext_vector_type(4)));

    // stages inside a chunk (index bits < LV), layout A
    static_for<0, (LOG2D < LV ? LOG2D : LV)>([&](auto s_) {
        constexpr int H = 1 << decltype(s_)::value;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (H >= 2) {
#pragma unroll
                for (int c = 0; c < VEC; c += 2)
                    if ((c & H) == 0) bfly2(r[k][c], r[k][c + 1], r[k][c | H], r[k][(c | H) + 1]);
            } else {
#pragma unroll
                for (int c = 0; c < VEC; ++c)
                    if ((c & H) == 0) bfly(r[k][c], r[k][c | H]);
            }
        }
    });

    if constexpr (LOG2D > LV) {
        constexpr int NL = (LOG2D - LV) < 6 ? (LOG2D - LV) : 6;   // lane-bit stages of this row length
        // ---- A -> B
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int c = 0; c < VEC; c += 4) {
                vec4 v = {r[k][c], r[k][c + 1], r[k][c + 2], r[k][c + 3]};
                *reinterpret_cast<vec4 *>(slab + k * PITCH + lane * VEC + c) = v;
            }
        wave_lds_fence();
        A *col = slab + (lane / VEC) * PITCH + (lane % VEC);
        A m[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) m[j] = col[j * VEC];
        // ---- the former lane stages, now in registers
        static_for<0, NL>([&](auto b_) {
            constexpr int H = 1 << decltype(b_)::value;
            if constexpr (H >= 2) {
#pragma unroll
                for (int j = 0; j < 64; j += 2)
                    if ((j & H) == 0) bfly2(m[j], m[j + 1], m[j | H], m[(j | H) + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < 64; j += 2) bfly(m[j], m[j + 1]);
            }
        });
        // ---- B -> A
#pragma unroll
        for (int j = 0; j < 64; ++j) col[j * VEC] = m[j];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int c = 0; c < VEC; c += 4) {
                vec4 v = *reinterpret_cast<const vec4 *>(slab + k * PITCH + lane * VEC + c);
                r[k][c] = v[0];
                r[k][c + 1] = v[1];
                r[k][c + 2] = v[2];
                r[k][c + 3] = v[3];
            }
        wave_lds_fence();   // the slab may be rewritten by this wave's next transform
    }

    // stages across chunks (index bits >= LV + 6), layout A
    static_for<LV + 6, (LOG2D > LV + 6 ? LOG2D : LV + 6)>([&](auto s_) {
        constexpr int KH = 1 << (decltype(s_)::value - LV - 6);
#pragma unroll
        for (int k = 0; k < K; ++k)
            if ((k & KH) == 0) {
#pragma unroll
                for (int c = 0; c < VEC; c += 2) bfly2(r[k][c], r[k][c + 1], r[k | KH][c], r[k | KH][c + 1]);
            }
    });
}

}  // namespace whvi
