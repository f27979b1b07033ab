"""WHVIRegression 3 -> 1024 -> 1024 -> 1 (BASELINE config 4's network) training step, batch 256, 1 MC sample:
eager vs whole-step hipGraph replay (1033 parameter tensors: the eager step is bound by per-parameter host work)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from whvi_amd.graphs import GraphedTrainStep
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = WHVIRegression([WHVILinear(3, 1024), nn.ReLU(), WHVILinear(1024, 1024), nn.ReLU(), WHVILinear(1024, 1)],
                     train_samples=1).to(dev).train()
if "philox" in sys.argv[1:]:
    net.set_inkernel_rng(True)
if "packed" in sys.argv[1:]:
    net.pack_parameters()          # 4 parameter tensors per stacked layer instead of 4 * stack
print(f"{len(list(net.parameters()))} parameter tensors", flush=True)
x, y = torch.randn(256, 3, device=dev), torch.randn(256, 1, device=dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)


def eager():
    opt.zero_grad(set_to_none=True)
    net.loss(x, y, n=45730).backward()
    opt.step()


for _ in range(15):
    eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    eager()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms per step", flush=True)
step = GraphedTrainStep(net, opt, x, y, n=45730)
l0 = float(step(x, y))
for _ in range(5):
    step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step(x, y)
torch.cuda.synchronize()
print(f"hipGraph replay: {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms per step; loss {l0:.1f} -> {float(step(x, y)):.1f}", flush=True)
