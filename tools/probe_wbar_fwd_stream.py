import sys, torch
sys.path.insert(0, '/root/repo')
from whvi_amd import _hip
dev = torch.device('cuda', 0)
def timed(fn, iters=30, warm=40):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
print(_hip.LIB_PATH)
for D, S in ((2048, 64), (2048, 256), (4096, 32), (512, 2048)):
    s1, s2, u = torch.randn(1, D, device=dev), torch.randn(1, D, device=dev), torch.randn(1, S, D, device=dev)
    ms = timed(lambda: _hip.wbar_fwd(s1, u, s2, D))
    print(f"D={D} S={S}: {S*D*D*4/1e6/ms:7.1f} GB/s written  {_hip.last_kernel()[6:]}", flush=True)
