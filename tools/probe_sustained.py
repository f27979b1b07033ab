"""Sustained run of the headline workload (D = 4096 f32, 2^20 rows in place): does the rate hold for half a minute?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip
x = torch.zeros(1 << 20, 4096, device="cuda")
x[:, 0] = 1e-30
out = []
for chunk in range(12):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(500):
        _hip.fwht_rows(x, out=x)
    e.record()
    torch.cuda.synchronize()
    x.zero_()
    x[:, 0] = 1e-30
    out.append(32 * 1.073741824 * 500 / s.elapsed_time(e))
    print(f"launches {chunk * 500:5d}..{chunk * 500 + 499:5d}: {out[-1]:.3f} TB/s", flush=True)
