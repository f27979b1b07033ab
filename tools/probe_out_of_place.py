"""tools/probe_out_of_place.py -- the drop-in entry point `fwht_cuda.fwht(X)` returns a NEW tensor (src/fwht/cuda/fwht_cuda.cpp:11):
the plain transform out of place against in place, 4 GiB per buffer, HIP events, per D and storage type; plus the full
`fwht_cuda.fwht` call (allocation of the result included)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fwht_cuda
from whvi_amd import _hip

dev = torch.device("cuda", 0)


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for dtype, esize in ((torch.float32, 4), (torch.float16, 2), (torch.float64, 8)):
    for d in (512, 2048, 4096):
        rows = (1 << 32) // (d * esize)
        x = (torch.randn(rows, d, device=dev) * 2.0 ** -8).to(dtype)
        y = torch.empty_like(x)
        oop = timed(lambda: _hip.fwht_rows(x, out=y))
        k_oop = _hip.last_kernel()
        full = timed(lambda: fwht_cuda.fwht(x))
        z = x.clone()
        inp = timed(lambda: (_hip.fwht_rows(z, out=z), z.mul_(2.0 ** -(d.bit_length() // 2)))[0])   # includes the rescale pass
        inp_only = timed(lambda: _hip.fwht_rows(y, out=y), iters=3, warm=1)                        # few launches: stays finite
        gb = 2 * x.numel() * esize / 1e6
        print(f"{str(dtype)[6:]:8s} D={d:5d}: out of place {gb / oop:7.1f} GB/s | fwht_cuda.fwht(X) {gb / full:7.1f} | in place "
              f"{gb / inp_only:7.1f}   {k_oop[6:]}", flush=True)
        del x, y, z
