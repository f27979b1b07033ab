"""tools/stream_forms_r04.py -- what round 4's "one contract for zero" costs: the streaming launch of the plain transform, 4 GiB in
place, HIP events, for the shapes whose drop-in launch moved from the signed to the unsigned lane stages (f32 D = 512 .. 2048,
f64 D = 64 .. 2048): default (whvi_fwht_<dtype>) vs opt-in WHVI_FWHT_SIGNED_LANES.  Run it under another build with
`python tools/_tuning.py --run <lib.so> tools/stream_forms_r04.py` for an A/B of a code form."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

for dtype, log2d in ((torch.float32, 9), (torch.float32, 10), (torch.float32, 11), (torch.float32, 12), (torch.float64, 6),
                     (torch.float64, 9), (torch.float64, 11)):
    d = 1 << log2d
    esize = torch.empty(0, dtype=dtype).element_size()
    rows = (4 << 30) // (d * esize)
    x = (torch.randn(rows, d, device="cuda", dtype=torch.float32) * 0.25).to(dtype)
    line = f"{str(dtype)[6:]:8s} D=2^{log2d:2d}:"
    for signed in (False, True, False, True):
        times = []
        for it in range(12):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            _hip.fwht_rows(x, out=x, signed_lanes=signed)
            e.record()
            torch.cuda.synchronize()
            times.append(s.elapsed_time(e))
            x.mul_(2.0 ** (-(log2d // 2) - (it & 1) * (log2d & 1)))
        ms = sum(times[4:]) / len(times[4:])
        line += f"  {'signed ' if signed else 'default'} {2 * x.numel() * esize / ms / 1e9:.3f} TB/s"
    print(line + "   " + _hip.last_kernel()[6:], flush=True)
    del x
