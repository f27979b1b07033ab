"""hipGraph replay of the launch-bound predictive pass.

A small WHVI network's Monte-Carlo forward is a few dozen tiny kernels (per layer: randn, softplus, the
fused weight launch, a batched GEMM, ...), i.e. bound by launch latency, not by the GPU.  The C ABI is
capture-safe (no allocation, no synchronisation, launches on the stream it is given), so the whole batched
pass can be recorded once into a hipGraph (``torch.cuda.graph``) and replayed: one launch per forward.
torch's generator is graph-aware, so every replay draws fresh eps.

Inference only: capturing backward + optimizer steps crashed the process with this torch/ROCm pairing when
tried (tools/toy_train.py), so training stays eager.
"""
import torch

__all__ = ["GraphedPredictor"]


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
