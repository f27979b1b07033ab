"""Sustained run of the headline workload (D = 4096 f32, 2^20 rows in place) on FINITE RANDOM data: does the rate hold for
half a minute?  An in-place transform multiplies the data's magnitude by 64, so every launch is followed by an exact 2^-6
rescale; each transform has its own HIP-event pair, the rescale is in neither the count nor the time.  (Round 1 ran this on
an all-zero buffer, which the chip clocks ~0.5 % higher: tools/profile_plateau.py.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip
x = torch.randn(1 << 20, 4096, device="cuda")
for chunk in range(12):
    pairs = []
    for _ in range(250):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        _hip.fwht_rows(x, out=x)
        e.record()
        x.mul_(2.0 ** -6)
        pairs.append((s, e))
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
    print(f"launches {chunk * 250:5d}..{chunk * 250 + 249:5d}: {32 * 1.073741824 / ms:.3f} TB/s  "
          f"finite={bool(torch.isfinite(x[::4099]).all())} max|x|={float(x[::4099].abs().max()):.2f}", flush=True)
