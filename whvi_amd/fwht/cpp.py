"""Host front-end (mirror of src/fwht/cpp/fwht.py:7-30) on the native OpenMP library."""
import torch.nn as nn
from torch.autograd import Function

import fwht_cpp

__all__ = ["FWHTFunction", "FWHT"]


class FWHTFunction(Function):
    """Batched FWHT along dimension 1 of a host tensor (src/fwht/cpp/fwht.py:7-18)."""

    @staticmethod
    def forward(ctx, x):
        return fwht_cpp.forward(x)

    @staticmethod
    def backward(ctx, grad_output):
        return fwht_cpp.backward(grad_output)


class FWHT(nn.Module):
    """Module wrapper (src/fwht/cpp/fwht.py:21-30)."""

    def forward(self, x):
        return FWHTFunction.apply(x)
