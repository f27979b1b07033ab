"""tools/profile_kernels.py -- the three headline kernels, a few launches each, for rocprofv3 --pmc runs.
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -- python3 tools/profile_kernels.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

dev = torch.device("cuda", 0)
x = torch.randn(1 << 18, 4096, device=dev) * 2.0 ** -40          # 4 GiB
for _ in range(3):
    _hip.fwht_rows(x, out=x)
del x
xh = (torch.randn(1 << 19, 4096, device=dev) * 2.0 ** -8).half()   # 4 GiB
for _ in range(3):
    _hip.fwht_rows(xh, out=xh)
del xh
d, S, B = 2048, 64, 8192
x = torch.randn(B * S, d, device=dev)
a, c, g = torch.randn(d, device=dev) * 0.01, torch.randn(d, device=dev) * 0.01, torch.randn(S, d, device=dev)
for _ in range(3):
    _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=1, out=x)
torch.cuda.synchronize()
