"""tools/profile_wbar_fwd.py -- the weight construction (whvi_wbar_fwd) at cache-resident shapes, a few launches each,
for rocprofv3 --kernel-trace (WHVI_WBAR_FWD_TILES=big|small selects the tile size for the A/B)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402  (tools/_tuning.py: the environment switches exist in measurement builds only)
if os.environ.get("WHVI_WBAR_FWD_TILES") or os.environ.get("WHVI_WBAR_FWD_XCD"):      # an environment switch is set: it exists in the measurement build only
    _tuning.use()
import torch
from whvi_amd import _hip

dev = torch.device("cuda", 0)
shapes = ((256, 16, 4), (3, 16, 64), (1, 32, 512), (1, 16, 1024), (1, 4, 2048), (1, 128, 512), (1, 16, 2048), (1, 64, 2048))
if os.environ.get("WHVI_WBAR_FWD_SHAPES") == "stream":
    shapes = ((1, 16, 2048), (1, 64, 2048), (1, 256, 512), (1, 1024, 512), (1, 16, 4096), (1, 128, 2048))
for (J, S, D) in shapes:
    s1, s2, u = torch.randn(J, D, device=dev), torch.randn(J, D, device=dev), torch.randn(J, 1 + S, D, device=dev)
    base = _hip.wbar_fwd(s1, u, s2, D, first=0, count=1).view(J, D, D)
    for _ in range(25):
        out = _hip.wbar_fwd(s1, u, s2, D, base=base, first=1)
    torch.cuda.synchronize()
    print(J, S, D, _hip.last_kernel(), bool(torch.isfinite(out).all()), flush=True)
    for _ in range(25):                                     # what WHVILinear.forward_mc calls: one or two launches by size
        out = _hip.wbar_fwd_mean(s1, u, s2, D)
    torch.cuda.synchronize()
    print(J, S, D, "production mean + samples:", _hip.last_kernel(), flush=True)
    u2 = u[:, 1:].contiguous()
    for _ in range(25):                                     # without the mean matrix (direct sampling): a pure write stream
        out = _hip.wbar_fwd(s1, u2, s2, D)
    torch.cuda.synchronize()
