"""Round-by-round timing of the config-3 fused launch (is the first round after allocation slower?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

dev = torch.device("cuda", 0)
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if pre:                                   # mimic bench.py: a 16 GiB buffer lived (and was freed) before
    y = torch.empty(1 << 32, dtype=torch.float32, device=dev)
    y.normal_()
    del y
d, S, B = 2048, 64, 8192
x = torch.randn(B * S, d, device=dev)
a, c, g = torch.randn(d, device=dev) * 0.01, torch.randn(d, device=dev) * 0.01, torch.randn(S, d, device=dev)
gb = 2 * x.numel() * 4 / 1e9
for rnd in range(6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
    for i in range(12):
        ev[i].record()
        _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=1, out=x)
    ev[12].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(12)]
    print(f"pre={pre} round {rnd}: " + " ".join(f"{gb / t:.2f}" for t in ts) + f"   absmax {float(x.abs().max()):.2e}", flush=True)
    if rnd == 2:
        x.normal_()
