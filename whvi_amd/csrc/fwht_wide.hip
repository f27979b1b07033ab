// whvi_amd/csrc/fwht_wide.hip -- streaming launch of the f32 one-row tile of 128 data registers (D = 8192) at three
// waves per SIMD, compiled with -fno-slp-vectorize (Makefile): see launch_wide_stream in dispatch.hpp.
#include "dispatch.hpp"

namespace whvi {

template <typename T>
void launch_wide_stream(u32x4 *d, const u32x4 *s, int64_t n_chunks, int64_t n_tiles, hipStream_t st)
{
    constexpr int LOG2D = max_single_pass_log2d<T>();
    constexpr int K = pick_k<T, LOG2D>();
    static_assert(tile_vgprs<T, K>() == 128, "one row = 128 data registers per lane");
    note_launch<T>("fwht_rows_kernel", LOG2D, K, POLICY_DPP, false, true, 256, 1, false);
    hipLaunchKernelGGL((fwht_rows_kernel<T, LOG2D, K, POLICY_DPP, false, true, 256, 1, false>),
                       dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, d, s, n_chunks, n_tiles);
}

template void launch_wide_stream<float>(u32x4 *, const u32x4 *, int64_t, int64_t, hipStream_t);

}  // namespace whvi
