"""tools/probe_persistent.py -- the final streaming recipe (256-thread blocks, store barrier, nt loads, sc1+nt stores,
XCD-contiguous order) as a PERSISTENT grid of N blocks per CU against the production one-tile-per-wave launch;
D = 4096 f32, 16 GiB in place, interleaved.  whvi_fwht_ex variant = 6 | align 1 << 6 | blocks-per-CU << 8."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

rows = 1 << 20
x = torch.randn(rows, 4096, device="cuda") * 2.0 ** -100
ref = x[:4096].clone()
want = _hip.fwht_rows(ref)
variants = {"production": None, "persistent x3": 6 | (1 << 6) | (3 << 8), "persistent x4": 6 | (1 << 6) | (4 << 8),
            "persistent x8": 6 | (1 << 6) | (8 << 8), "persistent x16": 6 | (1 << 6) | (16 << 8)}
for name, v in variants.items():          # same bits first
    y = ref.clone()
    _hip.fwht_rows(y, out=y, variant=v)
    assert torch.equal(y, want), name
res = {k: [] for k in variants}
for rnd in range(4):
    for name, v in variants.items():
        x.normal_().mul_(2.0 ** -100)
        for _ in range(2):
            _hip.fwht_rows(x, out=x, variant=v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            _hip.fwht_rows(x, out=x, variant=v)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 6)
for name, ms in res.items():
    ms.sort()
    print(f"{name:16s} median {ms[len(ms) // 2]:.4f} ms = {2 * x.numel() * 4 / ms[len(ms) // 2] / 1e9:.3f} TB/s (min {ms[0]:.4f})", flush=True)
