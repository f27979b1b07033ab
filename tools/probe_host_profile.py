"""cProfile of the eager training step of config 4's network with packed parameters: where does the host time go?"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = WHVIRegression([WHVILinear(3, 1024), nn.ReLU(), WHVILinear(1024, 1024), nn.ReLU(), WHVILinear(1024, 1)],
                     train_samples=1).to(dev).train().pack_parameters()
x, y = torch.randn(256, 3, device=dev), torch.randn(256, 1, device=dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)


def step():
    opt.zero_grad(set_to_none=True)
    net.loss(x, y, n=45730).backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
