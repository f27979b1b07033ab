"""Where the fused kernel's time goes (config 3: D = 2048, 64 MC samples, batch 8192, 4 GiB in place): the full
pipeline, the same without scale vectors (2 transforms), with only the per-sample vector, and the plain FWHT."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

dev = torch.device("cuda", 0)
d, S, B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 64, 8192
rows = B * S * 2048 // d
x = torch.randn(rows, d, device=dev)
a, c, g = torch.randn(d, device=dev) * 0.01, torch.randn(d, device=dev) * 0.01, torch.randn(S, d, device=dev)
stride = rows // S


def timed(fn, n=10):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    x.normal_()
    return ev[0].elapsed_time(ev[1]) / n


gb = 2 * x.numel() * 4 / 1e9
cases = (
    ("plain FWHT", lambda: _hip.fwht_rows(x, out=x)),
    ("fused a,b,c", lambda: _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=stride, out=x)),
    ("fused a,b,c interleaved", lambda: _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=1, out=x)),
    ("fused no scales", lambda: _hip.fused_shs(x, None, None, None, axis="col", n_samples=S, sample_stride=stride, out=x)),
    ("fused b only", lambda: _hip.fused_shs(x, None, g, None, axis="col", n_samples=S, sample_stride=stride, out=x)),
    ("fused a,c only", lambda: _hip.fused_shs(x, a, None, c, axis="col", n_samples=S, sample_stride=stride, out=x)),
    ("fused row-axis a,b,c", lambda: _hip.fused_shs(x, a[:1].contiguous(), g[:, :1].contiguous(), c[:1].contiguous(),
                                                    axis="row", n_samples=S, sample_stride=stride, group_rows=1, out=x)))
res = {name: [] for name, _ in cases}
for _ in range(5):                       # interleaved rounds: drift and bimodal phases show up as spread
    for name, fn in cases:
        res[name].append(timed(fn))
for name, _ in cases:
    v = sorted(res[name])
    print(f"D={d} {name:24s} median {v[2]:.3f} ms {gb / v[2]:.2f} TB/s   (min {gb / v[-1]:.2f} max {gb / v[0]:.2f})", flush=True)
