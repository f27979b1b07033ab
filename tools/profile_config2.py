"""tools/profile_config2.py -- BASELINE config 2 (WHVILinear(512, 512), 32 MC samples, batch 4096, fp32) for a rocprofv3
kernel trace: 30 batched forward + KL passes, then 30 forward + KL + backward passes, separated by a marker kernel
(torch.zeros(7)) so the two phases can be told apart in the trace.
    bash tools/prof_cmd.sh config2 tools/profile_config2.py && python tools/filter_stats.py gpurun_out/config2/config2_kernel_stats.csv"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd.layers import WHVILinear

dev = torch.device("cuda", 0)
layer = WHVILinear(512, 512).to(dev)
if len(sys.argv) > 1 and sys.argv[1] == "faithful":      # weight construction + GEMM instead of the shipped one-launch route
    layer.weight_submodule.faithful_dataflow = True
h = torch.randn(4096, 512, device=dev)
for phase in ("fwd", "train"):
    for it in range(40):
        if phase == "fwd":
            with torch.no_grad():
                out, kl = layer.forward_mc(h, 32), layer.kl           # exactly bench.py's `batched()`: no consumer op
        else:
            layer.zero_grad(set_to_none=True)
            (layer.forward_mc(h, 32).square().mean() + layer.kl).backward()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for it in range(20):
        if phase == "fwd":
            with torch.no_grad():
                out, kl = layer.forward_mc(h, 32), layer.kl           # exactly bench.py's `batched()`: no consumer op
        else:
            layer.zero_grad(set_to_none=True)
            (layer.forward_mc(h, 32).square().mean() + layer.kl).backward()
    e.record()
    torch.cuda.synchronize()
    print(phase, f"{s.elapsed_time(e) / 20:.3f} ms per pass", flush=True)
