"""tools/probe_wbar_mean.py -- `w_bar(g_mu) + w_bar(g_sigma eps_k)` as ONE launch computing both terms (whvi_wbar_fwd_mean)
against the mean matrix once + every sample with the mean added in its epilogue (two launches of whvi_wbar_fwd), interleaved,
HIP-event timed, at BASELINE config 2's and config 4's layer shapes and across the cache-resident / streaming boundary; and
config 2's forward + KL (WHVILinear(512, 512), 32 MC samples, batch 4096) with either form."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip
from whvi_amd.layers import WHVILinear

dev = torch.device("cuda", 0)


def timed(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for J, S, D in ((1, 32, 512), (256, 16, 4), (1, 16, 1024), (1, 128, 512), (1, 64, 1024), (1, 16, 4096), (1, 64, 2048), (1, 32, 4096)):
    s1, s2, u = torch.randn(J, D, device=dev), torch.randn(J, D, device=dev), torch.randn(J, 1 + S, D, device=dev)
    res = {}
    for rnd in range(3):
        for name, inline in (("one launch", True), ("two launches", False)):
            res.setdefault(name, []).append(timed(lambda: _hip.wbar_fwd_mean(s1, u, s2, D, inline=inline), 100, 30))
    mib = J * S * D * D * 4 / 2 ** 20
    print(f"J={J:3d} S={S:3d} D={D:4d} ({mib:7.1f} MiB): one launch {sorted(res['one launch'])[1]:8.2f} us   two launches "
          f"{sorted(res['two launches'])[1]:8.2f} us", flush=True)

layer = WHVILinear(512, 512).to(dev)
x = torch.randn(4096, 512, device=dev)
for name, limit in (("one launch", 256 << 20), ("two launches", 0)):
    _hip.WBAR_INLINE_MEAN_MAX_BYTES = limit
    def fwd():
        with torch.no_grad():
            out = layer.forward_mc(x, 32)
            return out, layer.kl
    print(f"config 2 forward + KL, weights by {name}: {timed(fwd, 50, 20) / 1e3:.4f} ms", flush=True)
