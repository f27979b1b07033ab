"""tools/probe_f16_data.py -- is the fp16 stream's rate data dependent?  D = 4096, 2^20 rows (8 GiB) in place, the
production launch, on: finite random data (re-scaled by 2^-6 after every launch so it stays in range), all zeros, and
data left to overflow to inf / NaN (what 20 un-rescaled in-place launches do to fp16, x 64 per transform)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

dev = torch.device("cuda", 0)
x = torch.empty(1 << 20, 4096, device=dev, dtype=torch.float16)
blk = torch.randn(4096, 4096, device=dev)


def each(iters, between):
    pairs = []
    for i in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _hip.fwht_rows(x, out=x)
        e1.record()
        between()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in pairs)
    return ts[len(ts) // 2]


def report(name, ms):
    print(f"{name:52s} {ms:.4f} ms  {2 * x.numel() * 2 / ms / 1e6:7.1f} GB/s  finite={bool(torch.isfinite(x[::4099].float()).all())}", flush=True)


for rnd in range(2):
    x.view(256, 4096, 4096).copy_((blk * 2.0 ** -8).half())
    each(12, lambda: x.mul_(2.0 ** -6))
    report("finite random, x 2^-6 between launches (per-launch events)", each(12, lambda: x.mul_(2.0 ** -6)))
    x.zero_()
    each(12, lambda: None)
    report("zeros, back to back (per-launch events)", each(12, lambda: None))
    report("zeros, x 2^-6 between launches (per-launch events)", each(12, lambda: x.mul_(2.0 ** -6)))
    x.view(256, 4096, 4096).copy_((blk * 2.0 ** -8).half())
    each(12, lambda: None)
    report("overflowed (inf / NaN), back to back", each(12, lambda: None))
    report("overflowed (inf / NaN), x 2^-6 between launches", each(12, lambda: x.mul_(2.0 ** -6)))
