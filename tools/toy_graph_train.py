"""tools/toy_graph_train.py -- the reference's toy-regression training step (3 WHVI layers 1 -> 128 -> 128 -> 1,
100 points, 1 MC sample, Adam, with KL) replayed from a hipGraph (whvi_amd.graphs.GraphedTrainStep).  Prints one
JSON line.  bench.py runs it as a child process so that this optional number can never take the bench line down."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from whvi_amd.graphs import GraphedTrainStep
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = WHVIRegression([WHVILinear(1, 128), nn.ReLU(), WHVILinear(128, 128), nn.ReLU(), WHVILinear(128, 1)],
                     train_samples=1).to(dev).train()
x = torch.linspace(-2, 2, 100, device=dev).unsqueeze(1)
y = torch.sin(3 * x)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, fused=True)     # one launch (the foreach form: 42)
step = GraphedTrainStep(net, opt, x, y, n=100)
first = float(step(x, y))
for _ in range(50):
    step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 2000
for _ in range(N):
    step(x, y)
torch.cuda.synchronize()
rate = N / (time.perf_counter() - t0)
print(json.dumps({"it_per_s": round(rate, 1), "first_loss": first, "last_loss": float(step(x, y))}), flush=True)
