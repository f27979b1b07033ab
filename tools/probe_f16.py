"""fp16 / bf16 D = 4096, 2^20 rows in place: production launch vs the tuning variants (whvi_fwht_ex; the 16-bit
translation units must be built with TUNABLE = true for variants other than 0 / 1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float16
log2d = int(sys.argv[2]) if len(sys.argv) > 2 else 12
x = (torch.randn(1 << (32 - log2d), 1 << log2d, device="cuda") * 2.0 ** -8).to(dt)


def bench(fn, iters=8, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


cases = {"prod": None, "dpp nt 256": 6, "dpp nt 256 barrier": 6 | (1 << 6), "lds nt 256": 7, "lds nt 256 barrier": 7 | (1 << 6),
         "lds nt 512": 7 | (1 << 4)}
res = {k: [] for k in cases}
for _ in range(3):
    for k, v in cases.items():
        res[k].append(bench(lambda: _hip.fwht_rows(x, out=x, variant=v)))
for k, v in res.items():
    ms = sorted(v)[1]
    print(f"{dt} D=2^{log2d} {k:20s}: {ms:.3f} ms {x.numel() * 4 / ms / 1e9:.2f} TB/s", flush=True)
