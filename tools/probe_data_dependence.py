"""Does the stream rate depend on the DATA (power / clocks)?  D = 4096, 4 GiB in place: zeros vs random, f32 and i32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip
d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows = (1 << 30) // d
for name, make in (("f32 zeros", lambda: torch.zeros(rows, d, device="cuda")),
                   ("f32 randn", lambda: torch.randn(rows, d, device="cuda") * 1e-3),
                   ("i32 zeros", lambda: torch.zeros(rows, d, device="cuda", dtype=torch.int32)),
                   ("i32 random", lambda: torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, d), device="cuda", dtype=torch.int32)),
                   ("f32 zeros again", lambda: torch.zeros(rows, d, device="cuda"))):
    x = make()
    keep = x.clone()
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
        x.copy_(keep)
    ts.sort()
    print(f"{name:16s}: {ts[1]:.4f} ms  {2 * (1 << 32) / ts[1] / 1e9:.2f} TB/s", flush=True)
    del x, keep
