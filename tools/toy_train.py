"""tools/toy_train.py -- the reference's toy-regression training loop (experiments/Toy example.ipynb:
3 WHVI layers 1 -> 128 -> 128 -> 1, ~100 points, 1 MC sample, Adam) on the GPU: eager iterations/s
and, when hipGraph capture of the whole step works, graph-replay iterations/s."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = WHVIRegression([WHVILinear(1, 128), nn.ReLU(), WHVILinear(128, 128), nn.ReLU(), WHVILinear(128, 1)],
                     train_samples=1).to(dev)
x = torch.linspace(-2, 2, 100, device=dev).unsqueeze(1)
y = torch.sin(3 * x) + 0.1 * torch.randn_like(x)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
net.train()


def step():
    opt.zero_grad(set_to_none=False)
    loss = net.loss(x, y, n=100)
    loss.backward()
    opt.step()
    return loss


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 200
for _ in range(N):
    step()
torch.cuda.synchronize()
eager = N / (time.perf_counter() - t0)
print(f"eager: {eager:.1f} it/s (with KL)", flush=True)

# The same step replayed from a hipGraph: tools/toy_graph_train.py (whvi_amd.graphs.GraphedTrainStep).  The process
# abort seen early in round 1 was a stale autograd graph (a kept `loss`) whose gradient accumulators lived on the
# default stream; GraphedTrainStep now detects that during its warm-up and raises RuntimeError instead.
