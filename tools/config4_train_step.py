"""tools/config4_train_step.py -- BASELINE config 4's network (WHVIRegression 3 -> 1024 -> 1024 -> 1) under the REFERENCE'S
training recipe: ``train_model`` (src/networks.py:71-99: two phases, ``scheduler.step()`` after every batch) with
``make_optimizer``'s Adam + decaying schedule (src/evaluation.py:15-27), batch 256, 1 MC sample (the reference's
train_samples default), with KL.  Three ways, ms per optimisation step over the same number of steps:

    eager    the reference's loop as it is (1033 parameter tensors: bound by per-parameter host work)
    packed   the same loop with the packed parameter layout (13 tensors; checkpoints keep the reference's keys)
    graphed  ``train_model(graphed=True)`` on the packed layout: every step ONE hipGraph replay holding loss, backward,
             Adam and the schedule (learning rate and step counter in device memory)

Prints one JSON line; bench.py runs it as a child process (``extras.config4_train_step``)."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from torch.utils.data import TensorDataset
from whvi_amd.evaluation import make_optimizer
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

dev = torch.device("cuda", 0)
BATCH, BATCHES = 256, 8
EPOCHS1, EPOCHS2 = (3, 9) if "--small" not in sys.argv else (1, 2)


def build():
    torch.manual_seed(0)
    return WHVIRegression([WHVILinear(3, 1024), nn.ReLU(), WHVILinear(1024, 1024), nn.ReLU(), WHVILinear(1024, 1)],
                          train_samples=1).to(dev)


g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn(BATCH * BATCHES, 3, device=dev, generator=g)
Y = torch.sin(X.sum(dim=1, keepdim=True))


class Batches:
    """What train_model needs of a data loader -- iteration over (x, y) batches and ``len(loader.dataset)`` -- without
    torch's per-sample collation (a DataLoader over a GPU TensorDataset spends 1.7 ms per batch of 256 slicing and
    stacking single rows: that would be the number measured)."""

    def __init__(self, x, y, batch):
        self.dataset = TensorDataset(x, y)
        self.batches = [(x[i:i + batch], y[i:i + batch]) for i in range(0, x.size(0), batch)]

    def __iter__(self):
        return iter(self.batches)


loader = Batches(X, Y, BATCH)
out = {"batch": BATCH, "mc_samples": 1, "steps_timed": BATCHES * EPOCHS2}
for name, kw_opt, kw_train in (("eager", {}, {}), ("packed", {"packed": True}, {"packed": True}),
                               ("graphed", {"packed": True, "capturable": True}, {"packed": True, "graphed": True})):
    net = build()
    # lambda0 = 0.03: an effective rate of ~1e-3 through the recipe's lambda0 ** 2 quirk, so the parameters visibly move
    optimizer, scheduler = make_optimizer(net, lambda0=0.03, **kw_opt)
    n_tensors = len(list(net.parameters()))
    net.train_model(loader, optimizer, scheduler, epochs1=EPOCHS1, epochs2=0, **kw_train)          # warm (and capture)
    l0 = float(net.current_mnll)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.train_model(loader, optimizer, scheduler, epochs1=0, epochs2=EPOCHS2, **kw_train)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / (BATCHES * EPOCHS2)
    finite = all(bool(torch.isfinite(p).all()) for p in net.parameters())
    out[name] = {"ms_per_step": round(ms, 3), "parameter_tensors": n_tensors, "mnll_first": round(l0, 3),
                 "mnll_last": round(float(net.current_mnll), 3), "values_finite": finite,
                 "lr_after": float(optimizer.param_groups[0]["lr"])}
print(json.dumps(out), flush=True)
