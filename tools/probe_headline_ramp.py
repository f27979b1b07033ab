"""Per-launch time of the headline workload from a cold start, as bench.py runs it (randn init, scale, W + K launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip
x = torch.randn(1 << 20, 4096, device="cuda")
x.mul_(2.0 ** -120)
n = 40
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
for i in range(n):
    ev[i].record()
    _hip.fwht_rows(x, out=x)
    if i == 19:
        x.mul_(2.0 ** -100)
ev[n].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(" ".join(f"{32 * 1.073741824 / t:.2f}" for t in ts), flush=True)
