"""tools/profile_config4.py [batch] -- BASELINE config 4's network (WHVIRegression 3 -> 1024 -> 1024 -> 1, 16 MC samples per
GPU) for a rocprofv3 kernel trace: 30 training-mode loss + backward passes at a small batch (default 2048: the GEMMs
shrink, what the framework itself launches stays).
    bash tools/prof_cmd.sh config4 tools/profile_config4.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

dev = torch.device("cuda", 0)
if len(sys.argv) > 1 and sys.argv[1] == "predict":
    # tools/profile_config4.py predict [faithful]: the PREDICTIVE pass bench.py times for config 4 (batch 45 730, 16 MC samples,
    # eval forward), 10 passes -- the kernel-time shares of the shipped route and of the as-written (faithful) one
    torch.manual_seed(0)
    net = WHVIRegression([WHVILinear(3, 1024), nn.ReLU(), WHVILinear(1024, 1024), nn.ReLU(), WHVILinear(1024, 1)],
                         eval_samples=16).to(dev).eval()
    if len(sys.argv) > 2 and sys.argv[2] == "faithful":
        net.set_faithful_dataflow(True)
    xb = torch.randn(45730, 3, device=dev)
    with torch.no_grad():
        for _ in range(3):
            net(xb)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            net(xb)
        e.record()
        torch.cuda.synchronize()
    print(f"predictive pass, batch 45730 x 16 MC: {s.elapsed_time(e) / 10:.3f} ms ({' '.join(sys.argv[2:]) or 'shipped route'})", flush=True)
    sys.exit(0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
net = WHVIRegression([WHVILinear(3, 1024), nn.ReLU(), WHVILinear(1024, 1024), nn.ReLU(), WHVILinear(1024, 1)],
                     train_samples=16, eval_samples=16).to(dev)
x, y = torch.randn(batch, 3, device=dev), torch.randn(batch, 1, device=dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
net.train()


def step():
    opt.zero_grad(set_to_none=True)
    net.loss(x, y, 45730).backward()
    opt.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(30):
    step()
e.record()
torch.cuda.synchronize()
print(f"batch {batch}: {s.elapsed_time(e) / 30:.3f} ms per training step (eager)", flush=True)
