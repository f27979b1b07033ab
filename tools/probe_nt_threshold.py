"""Cached 1024-thread launch vs the streaming launch (non-temporal, 256-thread blocks + store barrier) around the
Infinity Cache size, in place and out of place (D = 4096 f32): where should NT_MIN_BYTES sit?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

d = 4096
CACHED, STREAM = 2 | (2 << 4), 6 | (1 << 6)
for mib in (96, 128, 160, 192, 224, 256, 320, 384, 512, 768):
    rows = mib * (1 << 20) // (4 * d)
    x = torch.randn(rows, d, device="cuda") * 1e-30
    y = torch.empty_like(x)
    line = f"{mib:4d} MiB:"
    for place, out in (("in", x), ("oop", y)):
        for name, v in (("cached", CACHED), ("stream", STREAM)):
            best = 1e9
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    _hip.fwht_rows(x, out=out, variant=v)
                s.record()
                for _ in range(10):
                    _hip.fwht_rows(x, out=out, variant=v)
                e.record()
                torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / 10)
                x.mul_(1e-30).add_(1e-30)
            line += f"  {place}/{name} {2 * mib * 1.048576 / best / 1e3:5.2f}"
    print(line + "  TB/s", flush=True)
