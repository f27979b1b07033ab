#pragma once
// whvi_amd/csrc/tuning.hpp -- every switch that exists for MEASUREMENTS only, in one place.
//
// The shipped library (whvi_amd/csrc/Makefile never defines WHVI_TUNING_BUILD) contains none of them: no getenv, no
// alternative instruction forms, no trace hooks -- its launch form depends on its arguments alone
// (tests/test_abi.py::test_shipped_library_reads_no_environment).  Probe builds (`make -C whvi_amd/csrc tuning
// [DEFS=-D...]` -> whvi_amd/_exp/libwhvi_hip_tuning.so, loaded by tools/ through WHVI_HIP_LIB) define it and get
//   * WHVI_TUNE_ENV(name): environment A/B switches of the dispatch (read once per process), DESIGN.md 6.3;
//   * the -D overrides below, some of which produce WRONG VALUES on purpose (timing-only instruction swaps).
#ifdef WHVI_TUNING_BUILD
#include <stdlib.h>
#define WHVI_TUNE_ENV(name) getenv(name)
#else
#define WHVI_TUNE_ENV(name) ((const char *)nullptr)
#if defined(WHVI_F16_UNPACK) || defined(WHVI_F16_PACK_EXP) || defined(WHVI_BF16_PACK) || defined(WHVI_ROWS_WAVES_PER_EU) || \
    defined(WHVI_ROWS_PKMASK) || defined(WHVI_FUSED_PKMASK) || defined(WHVI_FUSED_SIGNED) || defined(WHVI_EXP_UNFUSED_DPP) || \
    defined(WHVI_NO_PK) || defined(WHVI_BLOCK_TRACE) || defined(WHVI_VEC_AUX) || defined(WHVI_FUSED_UPFRONT_8THS) || defined(WHVI_ROWS_BUFFER_IO) || defined(WHVI_FUSED_TILE_LOADS) || defined(WHVI_F64_STREAM_FORM) || defined(WHVI_FUSED_SHARED_PLAIN_NT) || defined(WHVI_WIDE_TILE_WAVES) || defined(WHVI_FUSED_TILE_LOADS_GROUPED) || defined(WHVI_VEC_LOAD_GROUPED) || defined(WHVI_ROWS_SETPRIO) || defined(WHVI_FUSED_SHARED_GLOBAL_STORES) || defined(WHVI_STORE_SPACING) || defined(WHVI_FUSED_STORE_FORM) || defined(WHVI_ROWS_LOAD_SPACING) || defined(WHVI_ROWS_STORE_FORM) || defined(WHVI_WIDE_TILE_LOADS) || defined(WHVI_ALIGN_SINGLE_PASS) || defined(WHVI_WBAR_FWD_STORE)
#error "kernel tuning switches need -DWHVI_TUNING_BUILD (make -C whvi_amd/csrc tuning DEFS=-D...)"
#endif
#endif

// ---- production values (a tuning build may override them with -D) ------------------------------------------------
#ifndef WHVI_F16_UNPACK
#define WHVI_F16_UNPACK 1          // fp16 unpack with an explicit shift for the high half (kernels.hpp: Elem<__half>)
#endif
#ifndef WHVI_BF16_PACK
#define WHVI_BF16_PACK 1           // one v_cvt_pk_bf16_f32 per output dword
#endif
#ifndef WHVI_ROWS_WAVES_PER_EU
#define WHVI_ROWS_WAVES_PER_EU 1   // minimum waves per SIMD fwht_rows_kernel is allocated for
#endif
#ifndef WHVI_ROWS_PKMASK
#define WHVI_ROWS_PKMASK 0         // explicit v_pk_add_f32 stages of the plain row kernel: none (fwht_tile.hpp)
#endif
#ifndef WHVI_FUSED_PKMASK
#define WHVI_FUSED_PKMASK 2        // fused kernel: packed adds in the permlane stages only (TU built with -fno-slp-vectorize)
#endif
#ifndef WHVI_FUSED_SIGNED
#define WHVI_FUSED_SIGNED 1        // fused kernel: signed DPP lane stages
#endif
#ifndef WHVI_VEC_AUX
#define WHVI_VEC_AUX 0             // cache-policy bits of the scale-vector loads that come from L2 (0 = default, cached in L1)
#endif
#ifndef WHVI_FUSED_UPFRONT_8THS
#define WHVI_FUSED_UPFRONT_8THS 0  // eighths of an L2-sourced scale vector requested before any of it is consumed (8 = all);
                                   // 0 = the per-shape choice of fused_shs_kernel
#endif
#ifndef WHVI_FUSED_TILE_LOADS
#define WHVI_FUSED_TILE_LOADS -1   // fused kernel, full tiles: -1 = per shape (kernels.hpp), 0 = bounds-checked buffer loads for every
                                   // tile (branch-free), 1 = global loads for full tiles behind a branch, 2 (tuning, WRONG on
                                   // ragged tails) = global loads without checks
#endif
#ifndef WHVI_F64_STREAM_FORM
#define WHVI_F64_STREAM_FORM 2     // f64 streams of 64-register tiles: 2 = 256-thread blocks + store barrier + signed (fma) DPP network
                                   // (production), 0 = 1024-thread blocks (round 2), 1 = 256 + barrier, unsigned; 3 = as 2 and
                                   // the signed network for 128-register tiles (D = 4096) too
#endif
#ifndef WHVI_FUSED_SHARED_PLAIN_NT
#define WHVI_FUSED_SHARED_PLAIN_NT 1   // fused kernel on a shared (cache-resident) source = a write-dominated stream: non-temporal
                                       // stores without the write-through bit (0: the sc1 nt stores of the read + write streams)
#endif
#ifndef WHVI_WIDE_TILE_WAVES
#define WHVI_WIDE_TILE_WAVES 3         // streaming launch of the f32 one-row tile of 128 data registers: waves per SIMD to compile for (0: the compiler's 2, in fwht_f32.hip)
#endif
#ifndef WHVI_ALIGN_SINGLE_PASS
#define WHVI_ALIGN_SINGLE_PASS -1      // store-barrier launches without the tile loop in the code: -1 = per-type rule (kernels.hpp), -2 = 16-bit storage only, 0 / 1 force
#endif
#ifndef WHVI_WIDE_TILE_LOADS
#define WHVI_WIDE_TILE_LOADS 0         // streaming launch of 128-register tiles: 1 = bounds-checked buffer loads from the wave-uniform tile base, 0 = global loads
#endif
#ifndef WHVI_ROWS_STORE_FORM
#define WHVI_ROWS_STORE_FORM 0         // streaming stores of the plain transform: 0 = one vector offset per chunk, 1 = scalar offsets, 2 = descending order (A/B)
#endif
#ifndef WHVI_FUSED_STORE_FORM
#define WHVI_FUSED_STORE_FORM 0        // fused kernel stores: 0 = chunk offset as the scalar offset, 1 = in the vector offset (A/B)
#endif
#ifndef WHVI_ROWS_LOAD_SPACING
#define WHVI_ROWS_LOAD_SPACING 0       // plain transform, full tiles: s_nop (N - 1) behind every tile load (A/B of issue spacing)
#endif
#ifndef WHVI_STORE_SPACING
#define WHVI_STORE_SPACING 1           // an s_nop between the streaming stores of the long-row block kernel (0: back to back; f32 D = 65536 6.05 -> 6.21-6.26 TB/s)
#endif
#ifndef WHVI_FUSED_SHARED_GLOBAL_STORES
#define WHVI_FUSED_SHARED_GLOBAL_STORES 0   // 1: fused kernel on a shared source, full tiles: back-to-back global nt stores.  MEASURED: one-transform launch 5.66 -> 5.25 TB/s written -- the spaced buffer stores stay
#endif
#ifndef WHVI_ROWS_SETPRIO
#define WHVI_ROWS_SETPRIO 0            // plain transform: 1 = s_setprio 3 around the tile loads, 2 = from the store barrier on (A/B)
#endif
#ifndef WHVI_VEC_LOAD_GROUPED
#define WHVI_VEC_LOAD_GROUPED -1       // fused kernel, L2-sourced scale vectors: four loads per scalar offset, issued back to back: -1 = f64 only, 0 / 1 force
#endif
#ifndef WHVI_FUSED_TILE_LOADS_GROUPED
#define WHVI_FUSED_TILE_LOADS_GROUPED 0   // fused kernel, tile loads through buffer instructions: four loads per scalar offset: -1 = f64 only, 0 / 1 force
#endif
