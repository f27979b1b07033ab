"""A/B of the production stream launch (256-thread blocks + store barrier) against 1024/512-thread blocks
(WHVI_STREAM_BIG_BLOCKS=1), per dtype; each setting in its own process (the switch is read once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402  (tools/_tuning.py: the environment switches exist in measurement builds only)
if os.environ.get("WHVI_STREAM_BIG_BLOCKS"):      # an environment switch is set: it exists in the measurement build only
    _tuning.use()
import torch
from whvi_amd import _hip
tag = "big-blocks" if os.environ.get("WHVI_STREAM_BIG_BLOCKS") else "256+barrier"
for dt, rows, log2d in ((torch.float32, 1 << 20, 12), (torch.float16, 1 << 20, 12), (torch.float64, 1 << 18, 11),
                        (torch.bfloat16, 1 << 20, 12), (torch.int32, 1 << 19, 10)):
    x = torch.zeros(rows, 1 << log2d, device="cuda", dtype=dt)
    x[:, ::7] = 1
    for _ in range(2): _hip.fwht_rows(x, out=x); x.zero_(); x[:, ::7] = 1
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); _hip.fwht_rows(x, out=x); _hip.fwht_rows(x, out=x); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 2); x.zero_(); x[:, ::7] = 1
    ts.sort()
    print(f"{tag:12s} {str(dt):15s} D=2^{log2d}: {ts[2]:.3f} ms {2 * x.numel() * x.element_size() / ts[2] / 1e9:.2f} TB/s", flush=True)
    del x
