"""Pure-torch front-ends (mirror of src/fwht/python/fwht.py): dense-matrix WHT and the
vectorised butterfly.  These are what the reference's model uses for HOST tensors
(src/weights.py:37-41), so they are kept op-for-op compatible: same operations in the same
order give the same bits."""
import torch
import torch.nn as nn
from torch.autograd import Function

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


class FWHTFunction(Function):
    """Vectorised batched FWHT (src/fwht/python/fwht.py:35-63).

    ``transform`` pairs ADJACENT elements first and doubles the trailing axis each round,
    which is the ascending-stride butterfly network: bit-equal to the C++/HIP transforms."""

    @staticmethod
    def transform(x: torch.Tensor) -> torch.Tensor:
        assert x.dim() == 2
        D = x.size(1)
        assert is_pow_of_2(D)
        y = x.unsqueeze(2)
        rounds = D.bit_length() - 1
        for _ in range(rounds):
            even, odd = y[:, ::2], y[:, 1::2]
            y = torch.cat((even + odd, even - odd), dim=2)
        return y.squeeze(1)

    @staticmethod
    def forward(ctx, x):
        return FWHTFunction.transform(x)

    @staticmethod
    def backward(ctx, grad_output):
        return FWHTFunction.transform(grad_output)


class FWHT(nn.Module):
    def forward(self, x):
        return FWHTFunction.apply(x)
