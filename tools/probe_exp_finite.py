"""tools/probe_exp_finite.py [f32|f16|bf16|i32] [log2d] [log2rows] -- interleaved A/B of the production library against
every whvi_amd/_exp/libexp_*.so (probe builds of kernels.hpp: make ... DEFS=-D...), in place, each timed launch on the SAME
finite data (an untimed copy from a master buffer) -- unlike tools/probe_exp.py, whose tiny constant data runs at a
different clock.  Defaults: the headline shape, f32 D = 4096, 2^20 rows."""
import ctypes
import glob
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

sfx = sys.argv[1] if len(sys.argv) > 1 else "f32"
log2d = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 20)
dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16, "i32": torch.int32}[sfx]
d = 1 << log2d
blk = min(rows, 4096)
seed = torch.randint(-99, 100, (blk, d), device="cuda", dtype=torch.int32) if sfx == "i32" else (torch.randn(blk, d, device="cuda") * 2.0 ** -8).to(dt)
master = seed.repeat(rows // blk, 1)
x = torch.empty_like(master)
libs = {"prod": _hip.lib()}
for path in sorted(glob.glob(os.path.join(os.path.dirname(_hip.LIB_PATH), "_exp", "libexp_*.so"))):
    libs[os.path.basename(path)[7:-3]] = ctypes.CDLL(path)
for L in libs.values():
    f = getattr(L, "whvi_fwht_" + sfx)
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {k: [] for k in libs}
for rnd in range(11):
    for k, L in libs.items():
        x.copy_(master)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(L, "whvi_fwht_" + sfx)(x.data_ptr(), x.data_ptr(), rows, log2d, st)
        e.record()
        torch.cuda.synchronize()
        assert rc == 0
        if rnd >= 2:
            res[k].append(s.elapsed_time(e))
for k, v in res.items():
    v.sort()
    med = v[len(v) // 2]
    print(f"{sfx} D=2^{log2d} {k:14s} median {med:.3f} ms  min {v[0]:.3f}  max {v[-1]:.3f}  -> {rows * 2 * d * x.element_size() / med / 1e9:.2f} TB/s", flush=True)
