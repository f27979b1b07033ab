import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from whvi_amd import _hip
x = (torch.randn(1 << 20, 4096, device="cuda") * 2.0 ** -8).half()
def bench(fn, iters=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for var, blks in ((0, (0,)), (6, (0, 1, 2)), (3, (0, 1, 2)), (7, (0, 1, 2))):
    for blk in blks:
        v = var | (blk << 4) if var else 0
        ms = bench(lambda: _hip.fwht_rows(x, out=x, variant=v if v else None))
        print(f"f16 D=4096 variant {var} blk {blk}: {ms:.3f} ms {x.numel()*4/ms/1e9:.2f} TB/s", flush=True)
