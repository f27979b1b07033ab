"""GPU front-end (mirror of src/fwht/cuda/fwht.py:5-16) on the MI355X HIP kernels."""
from torch.autograd import Function

import fwht_cuda

__all__ = ["FWHTFunction"]


class FWHTFunction(Function):
    """``FWHTFunction.apply(x)``: batched FWHT of the rows of a 2-D GPU tensor.

    The Walsh-Hadamard matrix is symmetric, so the backward pass is the same transform applied
    to the incoming gradient (src/fwht/cuda/fwht.py:14-16); going through ``apply`` again keeps
    it differentiable to any order, as in the reference."""

    @staticmethod
    def forward(ctx, x):
        return fwht_cuda.fwht(x)

    @staticmethod
    def backward(ctx, grad_output):
        return FWHTFunction.apply(grad_output)
