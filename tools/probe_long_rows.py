import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from whvi_amd import _hip
for log2d in (13, 14, 16, 18, 20):
    d = 1 << log2d
    rows = (1 << 30) // d
    x = torch.randn(rows, d, device="cuda") * 2.0 ** -60
    for _ in range(2): _hip.fwht_rows(x, out=x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): _hip.fwht_rows(x, out=x)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"D=2^{log2d}: {ms:.3f} ms for 4 GiB -> {2*x.numel()*4/ms/1e9:.2f} TB/s algorithmic", flush=True)
    del x
