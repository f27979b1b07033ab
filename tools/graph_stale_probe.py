"""tools/graph_stale_probe.py -- the failure GraphedTrainStep must turn into a Python error: a `loss` of an earlier
eager pass kept alive while the step is captured (its AccumulateGrad nodes belong to the default stream).  Expected
output: "RuntimeError raised" and then a successful capture once the tensor is dropped.  Run it on its own (one
process): if the check ever regresses, the capture aborts inside the HIP runtime."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from whvi_amd.graphs import GraphedTrainStep
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = WHVIRegression([nn.Linear(1, 32), nn.Tanh(), WHVILinear(32, 32, lambda_=1.0), nn.Tanh(), nn.Linear(32, 1)],
                     train_samples=2).to(dev).train()
x = torch.linspace(-1, 1, 64, device=dev).unsqueeze(1)
y = torch.sin(3 * x)
opt = torch.optim.Adam(net.parameters(), lr=1e-2, capturable=True)
kept = []
for _ in range(2):
    opt.zero_grad(set_to_none=False)
    loss = net.loss(x, y, n=64)
    loss.backward(retain_graph=True)
    kept.append(loss)                  # a list of losses: the autograd graphs stay alive
    opt.step()
torch.cuda.synchronize()
try:
    GraphedTrainStep(net, opt, x, y, n=64)
    print("NO ERROR (capture went through with a stale graph alive)", flush=True)
except RuntimeError as err:
    print("RuntimeError raised:", str(err)[:90], flush=True)
kept.clear()
del loss
step = GraphedTrainStep(net, opt, x, y, n=64)
print("capture after dropping the tensors ok, loss", float(step(x, y)), flush=True)
