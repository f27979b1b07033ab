"""Cache-resident sizes (D = 4096 f32, in place): block size of the cached launch vs the streaming launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

d = 4096
cases = {"prod": None, "cached 256": 2, "cached 512": 2 | (1 << 4), "cached 1024": 2 | (2 << 4), "stream 256+barrier": 6 | (1 << 6)}
for mib in (16, 32, 64, 128, 192, 256):
    rows = mib * (1 << 20) // (4 * d)
    x = torch.randn(rows, d, device="cuda") * 1e-30
    line = f"{mib:4d} MiB:"
    for name, v in cases.items():
        best = 1e9
        for _ in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                _hip.fwht_rows(x, out=x, variant=v)
            s.record()
            for _ in range(20):
                _hip.fwht_rows(x, out=x, variant=v)
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 20)
            x.mul_(1e-30).add_(1e-30)
        line += f"  {name} {2 * mib * 1.048576 / best / 1e3:5.2f}"
    print(line + "  TB/s", flush=True)
