"""tools/profile_wbar_bwd.py [iters] -- the one-launch backward of the weight construction (whvi_wbar_bwd) at the
shapes the BASELINE configs give it, a few launches each, HIP-event timed and suitable for rocprofv3 passes:

    rocprofv3 --kernel-trace --stats ... -- python3 tools/profile_wbar_bwd.py
    rocprofv3 --pmc FETCH_SIZE ...       -- python3 tools/profile_wbar_bwd.py 3

Shapes (J matrices x S samples of D x D gradients):  D = 2048 x 64 (1 GiB, the fused kernel's config-3 shape),
D = 512 x 32 (config 2's backward, 32 MiB), D = 1024 x 16 and D = 4 x 256 x 16 (config 4's layers), D = 4096 x 8."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
no_lds = os.environ.get("WHVI_WBAR_NO_LDS") == "1"        # A/B: the DPP network instead of the LDS-staged one
tiles = os.environ.get("WHVI_WBAR_TILES")                 # A/B: "small" / "big" (default: chosen by size)
dev = torch.device("cuda", 0)
shapes = ((1, 64, 2048, False), (1, 64, 2048, True), (1, 32, 512, True), (1, 16, 1024, True), (256, 16, 4, True),
          (1, 8, 4096, True), (1, 256, 2048, False))
if os.environ.get("WHVI_WBAR_SMALL_SHAPES") == "1":       # the cache-resident shapes only (tile-size A/B)
    shapes = ((1, 32, 512, True), (1, 16, 1024, True), (1, 4, 2048, True), (1, 128, 512, True), (1, 32, 1024, True),
              (256, 16, 4, True), (3, 16, 64, True), (1, 8, 128, False))
if os.environ.get("WHVI_WBAR_SMALL_SHAPES") == "2":       # around the tile-size crossover
    shapes = ((1, 16, 2048, True), (1, 256, 512, True), (1, 64, 1024, True), (1, 8, 4096, True), (1, 32, 2048, True),
              (1, 512, 512, True))
for (J, S, D, mean) in shapes:
    U = S + 1 if mean else S
    s1, s2 = torch.randn(J, D, device=dev), torch.randn(J, D, device=dev)
    u = torch.randn(J, U, D, device=dev)
    gw = torch.randn(J, S, D, D, device=dev)
    for _ in range(5):
        _hip.wbar_bwd(gw, s1, u, s2, mean=mean, no_lds=no_lds, tiles=tiles)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _hip.wbar_bwd(gw, s1, u, s2, mean=mean, no_lds=no_lds, tiles=tiles)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gb = gw.numel() * 4 / 1e9
    print(f"{'dpp' if no_lds else 'lds'} wbar_bwd J={J} S={S} D={D} mean={int(mean)}: {ms:.4f} ms, {gb / ms * 1e3:.0f} GB/s of dL/dW "
          f"({gb * 1e3:.0f} MB) = {gb / ms / 8:.3f} of 8 TB/s  {_hip.last_kernel()[6:]}", flush=True)
    del gw
