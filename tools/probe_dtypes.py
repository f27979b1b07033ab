"""Production launch per dtype: D = 4096, 4 GiB in place (steady state)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip
for dt in (torch.float32, torch.int32, torch.float64, torch.float16, torch.bfloat16):
    for d in (4096, 512):
        es = torch.empty((), dtype=dt).element_size()
        rows = (1 << 32) // (d * es)
        x = torch.zeros(rows, d, dtype=dt, device="cuda") if dt == torch.int32 else (torch.randn(rows, d, device="cuda") * 1e-3).to(dt)
        ts = []
        for rnd in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(10):
                _hip.fwht_rows(x, out=x)
            s.record()
            for _ in range(10):
                _hip.fwht_rows(x, out=x)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 10)
            if dt != torch.int32:
                x.mul_(0).add_(1e-3)
        ts.sort()
        print(f"{str(dt):15s} D={d:5d}: {ts[1]:.4f} ms  {2 * (1 << 32) / ts[1] / 1e9:.2f} TB/s  {rows / ts[1] / 1e6:.3f} Gtransforms/s", flush=True)
        del x
