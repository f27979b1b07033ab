"""The D sweep of the headline metric (4 GiB per D, f32, in place), steady state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip
for log2d in (6, 8, 9, 10, 11, 12):
    d = 1 << log2d
    x = torch.randn((1 << 30) // d, d, device="cuda") * 1e-30
    ts = []
    for rnd in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            _hip.fwht_rows(x, out=x)
        s.record()
        for _ in range(10):
            _hip.fwht_rows(x, out=x)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 10)
        x.mul_(1e-30).add_(1e-30)
    ts.sort()
    print(f"D={d:5d}: median {ts[1]:.4f} ms -> {x.numel() * 8 / ts[1] / 1e9:.2f} TB/s", flush=True)
