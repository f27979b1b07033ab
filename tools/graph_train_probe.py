"""tools/graph_train_probe.py -- where does whole-step hipGraph capture (forward + backward + Adam) fail?
Stages are flushed one by one; faulthandler prints the Python stack if the process dies."""
import faulthandler
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
import torch.nn as nn
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression


def say(*a):
    print(*a, flush=True)


dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
torch.manual_seed(0)
if mode == "plain":      # no WHVI layers at all: is it our code or torch/ROCm?
    net = nn.Sequential(nn.Linear(1, 128), nn.ReLU(), nn.Linear(128, 1)).to(dev)
    loss_fn = lambda: ((net(x) - y) ** 2).mean()
else:
    net = WHVIRegression([WHVILinear(1, 128), nn.ReLU(), WHVILinear(128, 128), nn.ReLU(), WHVILinear(128, 1)],
                         train_samples=1).to(dev).train()
    if mode == "loop":
        net.mc_mode = "loop"
    loss_fn = lambda: net.loss(x, y, n=100)
x = torch.linspace(-2, 2, 100, device=dev).unsqueeze(1)
y = torch.sin(3 * x)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        loss_fn().backward()
        opt.step()
torch.cuda.current_stream().wait_stream(s)
say(mode, "1 warm-up on side stream ok")
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
stage = os.environ.get("STAGE", "all")
with torch.cuda.graph(g):
    static_loss = loss_fn()
    say(mode, "2 forward captured")
    if stage in ("bwd", "all"):
        static_loss.backward()
        say(mode, "3 backward captured")
    if stage == "all":
        opt.step()
        say(mode, "4 optimizer captured")
say(mode, "5 capture closed")
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
say(mode, "6 replays ok, loss", float(static_loss))
