"""Production launch across problem sizes (D = 4096 f32, in place and out of place): where the launch-geometry
thresholds of dispatch.hpp (32 tiles per CU, 256 MiB non-temporal) sit relative to the measured curve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for mib in (1, 2, 4, 8, 16, 32, 64, 128, 192, 255, 256, 384, 512, 1024, 4096):
    rows = mib * (1 << 20) // (4 * d)
    x = torch.randn(rows, d, device="cuda") * 1e-30
    y = torch.empty_like(x)
    res = {}
    for name, fn in (("in place", lambda: _hip.fwht_rows(x, out=x)), ("out of place", lambda: _hip.fwht_rows(x, out=y))):
        best = 1e9
        for _ in range(3):
            n = max(4, min(200, int(2000 / max(mib, 1))))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn(); fn()
            s.record()
            for _ in range(n):
                fn()
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / n)
            x.mul_(1e-30).add_(1e-30)
        res[name] = best
    print(f"D={d} {mib:5d} MiB: " + "  ".join(f"{k} {v * 1e3:8.1f} us {2 * mib * 1.048576 / v / 1e3:6.2f} TB/s" for k, v in res.items()), flush=True)
