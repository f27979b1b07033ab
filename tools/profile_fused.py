"""tools/profile_fused.py -- fused S.H.G.H.S kernel only (config 3), a few launches, for rocprofv3 --pmc."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

dev = torch.device("cuda", 0)
d, S, B = 2048, 64, 8192
x = torch.randn(B * S, d, device=dev)
a, c, g = torch.randn(d, device=dev) * 0.01, torch.randn(d, device=dev) * 0.01, torch.randn(S, d, device=dev)
for _ in range(3):
    _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=1, out=x)
for _ in range(3):   # same traffic without the scale vectors: isolates their cost
    _hip.fused_shs(x, None, None, None, axis="col", n_samples=S, sample_stride=1, out=x)
torch.cuda.synchronize()
