"""tools/tune_fused.py -- time the column-axis fused kernel (config 3) for one WHVI_FUSED_TUNE setting
(the library reads the variable once per process, so each setting runs in its own process)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402  (tools/_tuning.py: the environment switches exist in measurement builds only)
if os.environ.get("WHVI_FUSED_TUNE"):      # an environment switch is set: it exists in the measurement build only
    _tuning.use()
import torch
from whvi_amd import _hip

d = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S, B = 64, 8192 * 2048 // d
dev = torch.device("cuda", 0)
x = torch.randn(B * S, d, device=dev)
a, c, g = torch.randn(d, device=dev) * 0.01, torch.randn(d, device=dev) * 0.01, torch.randn(S, d, device=dev)


def run():
    _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=1, out=x)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"TUNE={os.environ.get('WHVI_FUSED_TUNE', 'default')} D={d}: {ms:.3f} ms  {x.numel() * 8 / ms / 1e9:.2f} TB/s", flush=True)
