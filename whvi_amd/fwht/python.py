"""Pure-torch front-ends (mirror of src/fwht/python/fwht.py): dense-matrix WHT and the
vectorised butterfly.  These are what the reference's model uses for HOST tensors
(src/weights.py:37-41), so they are kept op-for-op compatible: same operations in the same
order give the same bits."""
import torch

from whvi_amd.fwht._frontends import make_fwht_function, make_fwht_module
from whvi_amd.utils import build_H, is_pow_of_2

__all__ = ["WHT_matmul", "FWHTFunction", "FWHT"]


class WHT_matmul:
    """Batched WHT as ``(H @ x.T).T`` with H built on first use and cached
    (src/fwht/python/fwht.py:9-32).  Differentiable through ordinary autograd."""

    def __init__(self):
        self.H = None
        self.H_built = False

    def apply(self, x: torch.Tensor) -> torch.Tensor:
        assert x.dim() == 2
        D = x.size(1)
        assert is_pow_of_2(D)
        if not self.H_built:
            self.H = build_H(D, x.device)
            self.H_built = True
        return (self.H @ x.T).T


def vectorised_fwht(x: torch.Tensor) -> torch.Tensor:
    """Batched FWHT of the rows of ``x`` with torch ops only (the algorithm of src/fwht/python/fwht.py:40-55).

    Every round pairs ADJACENT entries of the shrinking middle axis and appends sums then differences on the
    growing trailing axis; after log2(D) rounds the trailing axis is the transform.  That is the ascending-stride
    butterfly network, so the result is bit-equal to the native host and GPU transforms."""
    assert x.dim() == 2
    D = x.size(1)
    assert is_pow_of_2(D)
    work = x.unsqueeze(2)                                   # (batch, D, 1)
    for _ in range(D.bit_length() - 1):
        left, right = work[:, 0::2], work[:, 1::2]
        work = torch.cat((left + right, left - right), dim=2)
    return work.squeeze(1)                                  # (batch, 1, D) -> (batch, D)


FWHTFunction = make_fwht_function(
    vectorised_fwht, "FWHTFunction",
    "Vectorised batched FWHT in torch ops; ``FWHTFunction.transform(x)`` is the raw transform "
    "(src/fwht/python/fwht.py:35-63).")
FWHT = make_fwht_module(FWHTFunction, "Module form (src/fwht/python/fwht.py:66-74).")
