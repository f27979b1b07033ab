"""tools/graph_probe.py -- does hipGraph capture (torch.cuda.graph) work with the ctypes launches?  Step by
step, flushing after every stage so a crash shows where."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip


def say(*a):
    print(*a, flush=True)


dev = torch.device("cuda", 0)
x = torch.randint(-3, 4, (64, 512), device=dev).float()
y = torch.empty_like(x)
_hip.fwht_rows(x, out=y)
torch.cuda.synchronize()
say("1 eager ok")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    _hip.fwht_rows(x, out=y)
torch.cuda.current_stream().wait_stream(s)
say("2 side-stream warm ok")
with torch.cuda.graph(g):
    _hip.fwht_rows(x, out=y)
say("3 captured plain fwht")
y.zero_()
g.replay()
torch.cuda.synchronize()
say("4 replay ok:", bool(torch.equal(_hip.fwht_rows(y), x * 512)))

from whvi_amd.layers import WHVILinear
layer = WHVILinear(128, 128).to(dev)
h = torch.randn(100, 128, device=dev)
with torch.no_grad():
    for _ in range(3):
        out = layer.forward_mc(h, 4)
torch.cuda.synchronize()
say("5 eager forward_mc ok")
g2 = torch.cuda.CUDAGraph()
with torch.no_grad():
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = layer.forward_mc(h, 4)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g2):
        out = layer.forward_mc(h, 4)
say("6 captured forward_mc")
g2.replay()
torch.cuda.synchronize()
a = out.clone()
g2.replay()
torch.cuda.synchronize()
say("7 replays ok; fresh eps per replay:", not torch.equal(a, out), "finite:", bool(torch.isfinite(out).all()))
