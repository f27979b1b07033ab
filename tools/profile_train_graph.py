"""tools/profile_train_graph.py [steps] -- the captured training step of the UCI protocol's network (6 -> 128 -> 128 -> 1,
batch 64, 1 MC sample, packed, device-resident Adam + schedule) replayed `steps` times, for a rocprofv3 kernel trace: which
kernels one replay consists of and how much of its wall time they fill.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_graph -o tg -- python3 tools/profile_train_graph.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from whvi_amd.evaluation import make_optimizer
from whvi_amd.graphs import GraphedTrainStep
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = WHVIRegression([WHVILinear(6, 128, lambda_=3.0), nn.ReLU(), WHVILinear(128, 128, lambda_=3.0), nn.ReLU(),
                      WHVILinear(128, 1)], eval_samples=64).to(dev)
if os.environ.get("INKERNEL_RNG"):
    net.set_inkernel_rng()
opt, sched = make_optimizer(net, capturable=True, packed=True)
if os.environ.get("FUSED_ADAM"):
    from whvi_amd.evaluation import DeviceLambdaLR
    opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(1e-3, device=dev), capturable=True, fused=True)
    sched = DeviceLambdaLR(opt, lambda t: 1e-3 * torch.pow(1.0 + 0.0005 * t, -0.3), base_lrs=[1e-3])
if os.environ.get("NO_SCHED"):
    sched = None
x, y = torch.randn(64, 6, device=dev), torch.randn(64, 1, device=dev)
net.train()
step = GraphedTrainStep(net, opt, x, y, n=277, scheduler=sched)
for _ in range(50):
    step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(x, y)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{steps} replays: {dt / steps * 1e6:.1f} us per step (wall, incl. the two input copies)", flush=True)
t0 = time.perf_counter()
for _ in range(steps):
    step.graph.replay()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{steps} bare replays: {dt / steps * 1e6:.1f} us per step", flush=True)
