"""Does GraphedTrainStep capture after the network already trained eagerly on the default stream?  (child-process
probe: a failure here aborts inside the HIP runtime.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from whvi_amd.graphs import GraphedTrainStep
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = WHVIRegression([WHVILinear(1, 128), nn.ReLU(), WHVILinear(128, 128), nn.ReLU(), WHVILinear(128, 1)],
                     train_samples=1).to(dev).train()
x = torch.linspace(-2, 2, 100, device=dev).unsqueeze(1)
y = torch.sin(3 * x)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
for _ in range(5):                       # eager training on the default stream first
    opt.zero_grad(set_to_none=False)
    loss = net.loss(x, y, n=100)
    loss.backward()
    opt.step()
del loss
torch.cuda.synchronize()
print("eager steps done", flush=True)
step = GraphedTrainStep(net, opt, x, y, n=100)
print("captured", flush=True)
l0 = float(step(x, y))
for _ in range(300):
    step(x, y)
print("replayed", l0, float(step(x, y)), flush=True)
