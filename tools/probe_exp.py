"""tools/probe_exp.py -- interleaved A/B of experimental builds of the f32 FWHT kernel (whvi_amd/_exp/*.so,
built by hand with -DWHVI_EXP_*) against the production library, same process, same buffer."""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

LOG2D = int(sys.argv[1]) if len(sys.argv) > 1 else 12
SFX = sys.argv[2] if len(sys.argv) > 2 else "f32"           # f32 | f16 | bf16 | i32 | f64
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16, "i32": torch.int32, "f64": torch.float64}[SFX]
x = (torch.randn(1 << 30 if LOG2D < 12 else 1 << 32, device="cuda") * 2.0 ** -100).to(DT).view(-1, 1 << LOG2D)
libs = {"prod": _hip.lib()}
for path in sorted(glob.glob(os.path.join(os.path.dirname(_hip.LIB_PATH), "_exp", "libexp_*.so")) +
                   glob.glob(os.path.join(os.path.dirname(_hip.LIB_PATH), "_exp", "libwhvi_hip_ab_*.so"))):
    L = ctypes.CDLL(path)
    getattr(L, 'whvi_fwht_' + SFX).restype = ctypes.c_int
    getattr(L, 'whvi_fwht_' + SFX).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]
    libs[os.path.basename(path)[:-3].replace("libexp_", "").replace("libwhvi_hip_ab_", "")] = L
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ms(L, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    getattr(L, 'whvi_fwht_' + SFX)(x.data_ptr(), x.data_ptr(), x.size(0), LOG2D, st)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        getattr(L, 'whvi_fwht_' + SFX)(x.data_ptr(), x.data_ptr(), x.size(0), LOG2D, st)
    e.record()
    torch.cuda.synchronize()
    x.mul_(0).add_(1e-30)
    return s.elapsed_time(e) / iters


res = {k: [] for k in libs}
for rnd in range(5):
    for k, L in libs.items():
        res[k].append(ms(L))
for k, v in res.items():
    v.sort()
    print(f"{k:6s} median {v[len(v)//2]:.3f} ms  min {v[0]:.3f} ms  -> {x.numel()*2*x.element_size()/v[len(v)//2]/1e9:.2f} TB/s", flush=True)
