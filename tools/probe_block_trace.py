"""tools/probe_block_trace.py [log2d] [dtype] -- per-row timeline of fwht_block_rows_kernel: when wave 0 of the block
started on the row, had its tile transformed, finished the LDS exchange and issued its stores, and on which CU it ran.
Needs the traced build of the library (the production kernels carry no stamps); tools/_tuning.py builds it
(make -C whvi_amd/csrc tuning TAG=trace DEFS=-DWHVI_BLOCK_TRACE -> whvi_amd/_exp/libwhvi_hip_trace.so) and loads it:
    python tools/probe_block_trace.py 16 float32 [0|1]      (third argument: one row per block / pipelined grid)
Prints the average phase lengths and, per CU, the gap between one row's last stamp and the next row's first."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402  (tools/_tuning.py: the environment switches exist in measurement builds only)
_tuning.use("trace", "-DWHVI_BLOCK_TRACE")
import numpy as np
import torch

log2d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dtype = getattr(torch, sys.argv[2]) if len(sys.argv) > 2 else torch.float32
d = 1 << log2d
esize = torch.empty(0, dtype=dtype).element_size()
rows = (4 << 30) // (d * esize)
trace = torch.zeros(rows * 8, dtype=torch.int64, device="cuda")
os.environ["WHVI_BLOCK_TRACE"] = hex(trace.data_ptr())
from whvi_amd import _hip  # noqa: E402

if "trace" not in os.path.basename(_hip.LIB_PATH):
    sys.exit("the traced build was not loaded (see the docstring)")
VARIANT = {None: None, "0": 2 << 20, "1": 3 << 20}[sys.argv[3] if len(sys.argv) > 3 else None]
x = (torch.randn(rows, d, device="cuda") * 0.25).to(dtype)
for it in range(3):
    _hip.fwht_rows(x, out=x, variant=VARIANT)
    if dtype.is_floating_point:
        x.mul_(2.0 ** (-(log2d // 2)))
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
_hip.fwht_rows(x, out=x, variant=VARIANT)
e.record()
torch.cuda.synchronize()
print(f"{_hip.last_kernel()}  {s.elapsed_time(e):.3f} ms, {2 * x.numel() * esize / s.elapsed_time(e) / 1e9:.2f} TB/s (with the stamps)")
t = trace.cpu().numpy().reshape(rows, 8)
t0 = t[:, 0].min()
us = (t[:, :5] - t0) / 100.0                     # 100 MHz ticks -> microseconds
hw = t[:, 5].copy()
grid = int((hw != 0).sum())                     # persistent grids stamp the id for their first row only: row i ran on block i % grid
hw = hw[np.arange(rows) % grid] if grid < rows else hw
cu_key = hw & 0xFFFFFFFF
# HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ...; XCC_ID in the high word
cu = ((hw >> 32) & 0xF) * 4096 + ((cu_key >> 13) & 7) * 64 + ((cu_key >> 12) & 1) * 16 + ((cu_key >> 8) & 0xF)
print(f"rows {rows}, blocks {grid}, distinct CUs seen {len(np.unique(cu))}, span {us[:, 4].max():.1f} us")
names = ["start -> loads issued", "loads issued -> tile transformed", "tile transformed -> exchange done", "exchange done -> stores issued"]
for i, n in enumerate(names):
    dlt = us[:, i + 1] - us[:, i]
    print(f"  {n:36s} mean {dlt.mean():7.2f} us  p10 {np.percentile(dlt, 10):7.2f}  p90 {np.percentile(dlt, 90):7.2f}")
life = us[:, 4] - us[:, 0]
print(f"  block lifetime (start -> stores issued) mean {life.mean():.2f} us")
gaps, periods = [], []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    order = idx[np.argsort(us[idx, 0])]
    st, en = us[order, 0], us[order, 4]
    gaps.extend((st[1:] - en[:-1]).tolist())
    periods.extend((st[1:] - st[:-1]).tolist())
gaps, periods = np.array(gaps), np.array(periods)
print(f"  per CU: start-to-start period mean {periods.mean():.2f} us (p10 {np.percentile(periods, 10):.2f}, p90 {np.percentile(periods, 90):.2f}); "
      f"gap stores issued -> next block's start mean {gaps.mean():.2f} us (p10 {np.percentile(gaps, 10):.2f}, p90 {np.percentile(gaps, 90):.2f})")
