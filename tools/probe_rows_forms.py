"""tools/probe_rows_forms.py [GiB] -- the streaming launch of the plain transform per storage type and D (in place, HIP events):
TB/s and kernel symbol.  Run it under two builds (python tools/_tuning.py --run <lib.so> <script>) for an A/B of a launch / code form, e.g. the tile loop
of the store-barrier kernels run once (`make tuning TAG=sp DEFS=-DWHVI_ALIGN_SINGLE_PASS=1`) against the generic loop."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
for dtype in (torch.float32, torch.float16, torch.bfloat16, torch.int32, torch.float64):
    for log2d in (9, 11, 12, 13):
        if dtype == torch.float64 and log2d == 13:
            continue
        d = 1 << log2d
        esize = torch.empty(0, dtype=dtype).element_size()
        rows = int(gib * (1 << 30)) // (d * esize)
        x = torch.randint(-3, 4, (rows, d), device="cuda", dtype=torch.int32) if dtype == torch.int32 else \
            (torch.randn(rows, d, device="cuda", dtype=torch.float32) * 0.25).to(dtype)
        times = []
        for it in range(10):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            _hip.fwht_rows(x, out=x)
            e.record()
            torch.cuda.synchronize()
            times.append(s.elapsed_time(e))
            if dtype != torch.int32:
                x.mul_(2.0 ** (-(log2d // 2) - (it & 1) * (log2d & 1)))
        ms = sum(times[4:]) / len(times[4:])
        print(f"{str(dtype)[6:]:9s} D=2^{log2d:2d}: {2 * x.numel() * esize / ms / 1e9:.2f} TB/s  {_hip.last_kernel()[6:]}", flush=True)
        del x
