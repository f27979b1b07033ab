"""tools/profile_diag_apply.py [launches] -- whvi_diag_apply / whvi_diag_apply_bwd at BASELINE config 4's middle-layer shape
(D = 1024, 16 MC samples, batch 45 730: 3 GB read + 3 GB written; backward 6 GB read + 3 GB written) for rocprofv3 passes
(kernel trace; FETCH_SIZE / WRITE_SIZE one per pass): is the HBM traffic the algorithmic traffic?
    rocprofv3 --kernel-trace --stats -- python3 tools/profile_diag_apply.py 20"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
D, S, B = 1024, 16, 45730
g = torch.Generator(device=dev).manual_seed(0)
s1, s2, bias = (torch.randn(D, device=dev, generator=g) for _ in range(3))
u = torch.randn(1 + S, D, device=dev, generator=g)
x = torch.randn(S, B, D, device=dev, generator=g)
out = torch.empty_like(x)
gout = torch.randn(S, B, D, device=dev, generator=g)
for _ in range(n):
    _hip.diag_apply(x, s1, s2, u, bias, n_samples=S, out=out, relu_in=True, relu_out=True)
print("forward:", _hip.last_kernel(), "algorithmic bytes", 2 * x.numel() * 4, flush=True)
for _ in range(n):
    _hip.diag_apply_bwd(gout, x, s1, s2, u, n_samples=S, bias=bias, relu_in=True, relu_out=True)
print("backward:", _hip.last_kernel(), "algorithmic bytes", 3 * x.numel() * 4, flush=True)
torch.cuda.synchronize()
