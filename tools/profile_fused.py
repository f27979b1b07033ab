"""tools/profile_fused.py [launches] -- fused S.H.G.H.S kernel only (BASELINE config 3: D = 2048, 64 MC samples, batch 8192
= 4 GiB in place), for rocprofv3: `--kernel-trace --stats` with the default 40 launches (the clocks need ~25 to ramp),
`--pmc FETCH_SIZE` / `WRITE_SIZE` / SQ_* passes with 3.  The last 3 launches run WITHOUT the scale vectors: same data
traffic, isolates what the three vectors cost."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
d, S, B = 2048, 64, 8192
x = torch.randn(B * S, d, device=dev)
# S1, S2 = random signs / sqrt(D), g ~ N(0, 1): a launch preserves the data's norm in expectation (as in bench.py)
a = (torch.randint(0, 2, (d,), device=dev).float() * 2 - 1) * d ** -0.5
c = (torch.randint(0, 2, (d,), device=dev).float() * 2 - 1) * d ** -0.5
g = torch.randn(S, d, device=dev)
for _ in range(n):
    _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=1, out=x)
print(_hip.last_kernel(), "finite:", bool(torch.isfinite(x[::4099]).all()), flush=True)
x.mul_(2.0 ** -40)
for _ in range(3):   # H.H = D.I: each of these multiplies the data by 2048
    _hip.fused_shs(x, None, None, None, axis="col", n_samples=S, sample_stride=1, out=x)
torch.cuda.synchronize()
