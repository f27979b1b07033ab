"""Interleaved A/B of block sizes for the production-form f32 kernel (variant 6 = DPP, one tile per wave,
non-temporal loads + write-through stores) on the 16 GiB in-place workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip
LOG2D = int(sys.argv[1]) if len(sys.argv) > 1 else 12
x = (torch.randn(1 << 20, 4096, device="cuda") * 2.0 ** -120).view(-1, 1 << LOG2D)
def ms(v, iters=6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _hip.fwht_rows(x, out=x, variant=v); torch.cuda.synchronize(); s.record()
    for _ in range(iters): _hip.fwht_rows(x, out=x, variant=v)
    e.record(); torch.cuda.synchronize()
    x.mul_(2.0 ** -(7 * LOG2D // 2))     # 7 transforms grow by 2^(7*log2d/2)
    return s.elapsed_time(e) / iters
cfg = {"prod(0)": None}
for blk, name in ((3, 128), (0, 256)):
    for al in (0, 1, 2):
        cfg[f"B={name} align={al}"] = 6 | (blk << 4) | (al << 6)
res = {k: [] for k in cfg}
for _ in range(5):
    for k, v in cfg.items():
        res[k].append(ms(v))
for k, v in res.items():
    v.sort(); print(f"D=2^{LOG2D} {k:16s} median {v[2]:.3f} ms -> {x.numel()*8/v[2]/1e9:.2f} TB/s", flush=True)
