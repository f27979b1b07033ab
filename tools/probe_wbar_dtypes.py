import sys, torch
sys.path.insert(0, '/root/repo')
from whvi_amd import _hip
dev = torch.device('cuda', 0)
def timed(fn, iters=20, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for dtype in (torch.float32, torch.float64):
    es = 4 if dtype == torch.float32 else 8
    for D, S in ((512, 1024 * 4 // es), (2048, 256 // es), (4096, 64 // es)):
        s1, s2 = torch.randn(1, D, device=dev, dtype=dtype), torch.randn(1, D, device=dev, dtype=dtype)
        u = torch.randn(1, 1 + S, D, device=dev, dtype=dtype)
        gb = S * D * D * es / 1e6
        ms = timed(lambda: _hip.wbar_fwd_mean(s1, u, s2, D)); kf = _hip.last_kernel()
        gw = torch.randn(1, S, D, D, device=dev, dtype=dtype)
        msb = timed(lambda: _hip.wbar_bwd(gw, s1, u, s2, mean=True)); kb = _hip.last_kernel()
        print(f"{str(dtype)[6:]:8s} D={D:5d} S={S:4d} ({gb/1024:.2f} GiB): fwd(mean+samples) {gb/ms:7.1f} GB/s {kf[6:]} | bwd {gb/msb:7.1f} GB/s {kb[6:]}", flush=True)
        del gw
