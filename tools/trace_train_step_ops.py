"""tools/trace_train_step_ops.py -- which Python lines launch the glue kernels of one training step of the UCI protocol's
network (torch.profiler with stacks, eager, packed parameters + fused capturable Adam + one-launch schedule)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from torch.profiler import ProfilerActivity, profile
from whvi_amd.evaluation import make_optimizer
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = WHVIRegression([WHVILinear(6, 128, lambda_=3.0), nn.ReLU(), WHVILinear(128, 128, lambda_=3.0), nn.ReLU(),
                      WHVILinear(128, 1)], eval_samples=64).to(dev)
opt, sched = make_optimizer(net, capturable=True, packed=True)
x, y = torch.randn(64, 6, device=dev), torch.randn(64, 1, device=dev)
net.train()


def step():
    opt.zero_grad(set_to_none=True)
    net.loss(x, y, 277).backward()
    opt.step()
    sched.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.kernels:
        stack = [s for s in (ev.stack or []) if "/whvi_amd/" in s]
        rows.append((ev.name, len(ev.kernels), stack[0].split("/whvi_amd/")[-1] if stack else (ev.stack[0] if ev.stack else "?")))
for name, n, where in rows:
    print(f"{name:45s} {n}  {where[:110]}")
print("ops with kernels:", len(rows), " kernels:", sum(n for _, n, _ in rows))
