import sys, time, torch
sys.path.insert(0, '/root/repo')
import fwht_cuda
from whvi_amd import _hip
x = torch.randn(512, 512, device='cuda')
for fn, name in ((lambda: fwht_cuda.fwht(x), "fwht_cuda.fwht(512x512)"), (lambda: _hip.fwht_rows(x), "_hip.fwht_rows"), (lambda: torch.empty_like(x), "torch.empty_like"), (lambda: x + 1, "torch add")):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 5000
    for _ in range(N): fn()
    t1 = time.perf_counter()          # host-side issue cost (async)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:28s} host issue {1e6*(t1-t0)/N:6.2f} us/call   incl. drain {1e6*(t2-t0)/N:6.2f} us/call")
