"""tools/probe_block_rows.py -- fwht_block_rows_kernel at the row lengths it serves, 4 GiB in place, finite data.
argv[1] = 0 / 1 forces one row per block / the persistent pipelined grid (bits 20..22 of whvi_fwht_ex's variant word)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

pipe = sys.argv[1] if len(sys.argv) > 1 else "default"
variant = {"default": None, "0": 2 << 20, "1": 3 << 20}[pipe]
for dtype, log2d in ((torch.float32, 16), (torch.float32, 15), (torch.float32, 14), (torch.float16, 16), (torch.float64, 15),
                     (torch.int32, 16)):
    d = 1 << log2d
    esize = torch.empty(0, dtype=dtype).element_size()
    rows = (4 << 30) // (d * esize)
    if dtype == torch.int32:
        x = torch.randint(-3, 4, (rows, d), device="cuda", dtype=torch.int32)
    else:
        x = (torch.randn(rows, d, device="cuda", dtype=torch.float32) * 0.25).to(dtype)
    times = []
    for it in range(8):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        _hip.fwht_rows(x, out=x, variant=variant)
        e.record()
        torch.cuda.synchronize()
        times.append(s.elapsed_time(e))
        if dtype != torch.int32:
            x.mul_(2.0 ** (-(log2d // 2) - (it & 1) * (log2d & 1)))
    ms = sum(times[3:]) / len(times[3:])
    print(f"pipe={pipe} {str(dtype)[6:]:8s} D=2^{log2d}: {ms:.3f} ms -> "
          f"{2 * x.numel() * esize / ms / 1e9:.2f} TB/s  {_hip.last_kernel()[6:]}", flush=True)
    del x
