"""Per-call cost of the WHVILinear flavours at a small batch (host / launch overhead dominates): forward_mc and
forward + backward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd.layers import WHVILinear

dev = "cuda"


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for n_in, n_out in ((3, 1024), (1024, 1024), (1024, 1), (1, 128), (128, 128), (13, 128), (100, 300)):
    layer = WHVILinear(n_in, n_out).to(dev)
    x = torch.randn(64, n_in, device=dev)

    def fwd():
        with torch.no_grad():
            layer.forward_mc(x, 16)

    def fwd1():
        with torch.no_grad():
            layer(x)

    def step():
        layer.zero_grad(set_to_none=True)
        (layer.forward_mc(x, 16).square().mean() + layer.kl).backward()
    print(f"WHVILinear({n_in},{n_out}) [{type(layer.weight_submodule).__name__}]: forward {timed(fwd1):.3f} ms, forward_mc(16) {timed(fwd):.3f} ms, "
          f"forward_mc + kl + backward {timed(step):.3f} ms", flush=True)

import torch.nn as nn
from whvi_amd.networks import WHVIRegression
net = WHVIRegression([WHVILinear(3, 1024), nn.ReLU(), WHVILinear(1024, 1024), nn.ReLU(), WHVILinear(1024, 1)],
                     train_samples=1).to(dev).train()
xb, yb = torch.randn(256, 3, device=dev), torch.randn(256, 1, device=dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)


def train_step():
    opt.zero_grad(set_to_none=True)
    net.loss(xb, yb, n=45730).backward()
    opt.step()


def fwd_loss():
    with torch.no_grad():
        net.loss(xb, yb, n=45730)


print(f"WHVIRegression 3-1024-1024-1, batch 256, 1 MC sample: loss forward {timed(fwd_loss):.3f} ms, training step {timed(train_step, 10, 3):.3f} ms "
      f"({len(list(net.parameters()))} parameter tensors)", flush=True)
