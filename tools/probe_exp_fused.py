"""Interleaved A/B of experimental builds of the fused kernel (whvi_amd/_exp/libexpf_*.so, built by hand with
-DWHVI_FUSED_*) against the production library: config 3 (D = 2048) and D = 4096, 4 GiB in place."""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
libs = {"prod": _hip.lib()}
for path in sorted(glob.glob(os.path.join(os.path.dirname(_hip.LIB_PATH), "_exp", "libexpf_*.so"))):
    L = ctypes.CDLL(path)
    L.whvi_fused_shs_f32.restype = ctypes.c_int
    L.whvi_fused_shs_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, i64, i64, i64, i32, vp]
    libs[os.path.basename(path)[8:-3]] = L
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for log2d in (11, 12, 10):
    d, S = 1 << log2d, 64
    rows = (1 << 30) // d
    x = torch.randn(rows, d, device="cuda")
    a, c, g = torch.randn(d, device="cuda") * 0.01, torch.randn(d, device="cuda") * 0.01, torch.randn(S, d, device="cuda")

    def ms(L, iters=10):
        call = lambda: L.whvi_fused_shs_f32(x.data_ptr(), x.data_ptr(), a.data_ptr(), g.data_ptr(), c.data_ptr(), rows, log2d, S, 1, 1, 1, st)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(8):
            call()
        s.record()
        for _ in range(iters):
            call()
        e.record()
        torch.cuda.synchronize()
        x.normal_()
        return s.elapsed_time(e) / iters
    for L in libs.values():
        for _ in range(3):
            ms(L)                      # clock ramp
    res = {k: [] for k in libs}
    for rnd in range(5):
        for k, L in libs.items():
            res[k].append(ms(L))
    for k, v in res.items():
        v.sort()
        print(f"D={d} {k:6s} median {v[2]:.3f} ms -> {x.numel() * 8 / v[2] / 1e9:.2f} TB/s", flush=True)
