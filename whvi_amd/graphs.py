"""hipGraph replay of the launch-bound predictive pass.

A small WHVI network's Monte-Carlo forward is a few dozen tiny kernels (per layer: randn, softplus, the
fused weight launch, a batched GEMM, ...), i.e. bound by launch latency, not by the GPU.  The C ABI is
capture-safe (no allocation, no synchronisation, launches on the stream it is given), so the whole batched
pass can be recorded once into a hipGraph (``torch.cuda.graph``) and replayed: one launch per forward.
torch's generator is graph-aware, so every replay draws fresh eps.

``GraphedTrainStep`` does the same for a whole training step (loss, backward, optimizer): the backward of the
fused weight kernel runs from the autograd engine's thread and is captured with the rest
(tools/graph_train_probe.py walks through the stages).
"""
import gc
import warnings

import torch

__all__ = ["GraphedPredictor", "GraphedTrainStep"]


class GraphedPredictor:
    """``pred = GraphedPredictor(net, example_x, n_samples)(x)`` == ``net.forward_batched(x, n_samples)``
    under ``torch.no_grad()``, replayed from a hipGraph.  ``x`` must keep the example's shape and dtype; it
    is copied into a static buffer and the returned tensor is a static buffer too (clone it to keep it)."""

    def __init__(self, net, example_x: torch.Tensor, n_samples: int = None, warmup: int = 3):
        if example_x.device.type != "cuda":
            raise RuntimeError("GraphedPredictor needs a GPU tensor")
        self.net = net
        self.n_samples = int(n_samples if n_samples is not None else net.eval_samples)
        self.static_x = example_x.detach().clone()
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=example_x.device)
        with torch.no_grad():
            side.wait_stream(torch.cuda.current_stream(example_x.device))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    net.forward_batched(self.static_x, self.n_samples)
            torch.cuda.current_stream(example_x.device).wait_stream(side)
            with torch.cuda.graph(self.graph):
                self.static_out = net.forward_batched(self.static_x, self.n_samples)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self.static_x.shape or x.dtype != self.static_x.dtype:
            raise RuntimeError("GraphedPredictor: input shape/dtype differs from the captured example")
        self.static_x.copy_(x)
        self.graph.replay()
        return self.static_out


class GraphedTrainStep:
    """One optimisation step of a ``WHVINetwork`` -- ``loss(x, y, n)``, ``backward()``, ``optimizer.step()`` --
    recorded into a hipGraph and replayed: ``loss = step(x, y)``.

    The optimizer must keep its state on the device (``torch.optim.Adam(..., capturable=True)``); learning-rate
    schedules that change ``lr`` from the host are not captured.  ``x`` / ``y`` keep the example's shapes; the
    returned loss is a static tensor overwritten by the next call.  Fresh eps are drawn on every replay.

    It may be built after the network has trained eagerly, provided no tensor of an earlier pass that still carries
    an autograd graph (a kept ``loss``) is alive: gradient accumulators created by a backward pass on another
    stream would make the capture abort the PROCESS inside the HIP runtime.  That precondition is checked: the
    side-stream warm-up runs with torch's "AccumulateGrad node's stream does not match" warning promoted, and a
    ``RuntimeError`` is raised before capture begins when it fires.  ``WHVINetwork.loss`` itself keeps only
    detached monitoring values."""

    def __init__(self, net, optimizer, example_x, example_y, n: int, ignore_kl: bool = False, warmup: int = 3):
        if example_x.device.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs GPU tensors")
        for group in optimizer.param_groups:
            if "capturable" in group and not group["capturable"]:
                raise RuntimeError("GraphedTrainStep: create the optimizer with capturable=True")
        self.net, self.optimizer, self.n, self.ignore_kl = net, optimizer, int(n), bool(ignore_kl)
        self.static_x, self.static_y = example_x.detach().clone(), example_y.detach().clone()
        dev = example_x.device
        # gradient accumulators of an earlier eager backward belong to the stream that ran it; they die with the
        # last reference to that pass's autograd graph, so drop ours before warming up on the capture side stream
        net._pass_kl = None
        for module in net.modules():
            if hasattr(module, "_mc_kl"):
                module._mc_kl = None
        optimizer.zero_grad(set_to_none=True)
        gc.collect()
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # A stale autograd graph (a kept ``loss`` of an earlier eager pass) keeps the parameters' AccumulateGrad nodes
        # alive on the stream that created them; the warm-up's backward on the side stream then has to synchronise
        # with that stream -- harmless here, a process abort inside torch.cuda.graph.  torch warns about exactly that
        # (once per process unless warn-always is on), so the warm-up doubles as the check.
        warn_always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)
        try:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                with torch.cuda.stream(side):
                    for _ in range(max(1, warmup)):
                        self._eager_step()
        finally:
            torch.set_warn_always(warn_always)
        torch.cuda.current_stream(dev).wait_stream(side)
        stale = [w for w in caught if "AccumulateGrad node's stream" in str(w.message)]
        for w in caught:                             # everything else is passed on unchanged
            if w not in stale:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if stale:
            optimizer.zero_grad(set_to_none=True)
            raise RuntimeError(
                "GraphedTrainStep: an autograd graph of an earlier pass is still alive (a kept `loss` tensor, a list of "
                "losses, ...): its gradient accumulators belong to another stream and capturing a backward pass "
                "through them would abort inside the HIP runtime. Drop those tensors (keep `loss.detach()` or "
                "`float(loss)` instead) and build the GraphedTrainStep again.")
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.static_loss = net.loss(self.static_x, self.static_y, self.n, ignore_kl=self.ignore_kl)
            self.static_loss.backward()
            optimizer.step()

    def _eager_step(self):
        self.optimizer.zero_grad(set_to_none=True)
        self.net.loss(self.static_x, self.static_y, self.n, ignore_kl=self.ignore_kl).backward()
        self.optimizer.step()

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if x.shape != self.static_x.shape or y.shape != self.static_y.shape:
            raise RuntimeError("GraphedTrainStep: batch shape differs from the captured example")
        self.static_x.copy_(x)
        self.static_y.copy_(y)
        self.graph.replay()
        return self.static_loss.detach()
