"""Shared machinery of the three FWHT front-ends (GPU kernels, native host library, torch ops).

The Walsh-Hadamard matrix is symmetric, so for ``y = x . H`` the vector-Jacobian product of an incoming
gradient ``g`` is ``g . H`` -- the same transform again.  Every front-end is therefore the same
``autograd.Function`` around a different ``transform`` callable; routing the backward pass through ``apply``
keeps it differentiable to any order (what the reference's GPU front-end does, src/fwht/cuda/fwht.py:14-16;
its host front-ends stop at first order, src/fwht/cpp/fwht.py:16-18 and src/fwht/python/fwht.py:61-63).
"""
import torch.nn as nn
from torch.autograd import Function


def make_fwht_function(transform, name, doc):
    """Build the ``FWHTFunction`` class of one backend from its row-transform callable."""

    class _SelfAdjointTransform(Function):
        @staticmethod
        def forward(ctx, x):
            return transform(x)

        @staticmethod
        def backward(ctx, grad_output):
            return _SelfAdjointTransform.apply(grad_output)

    _SelfAdjointTransform.transform = staticmethod(transform)
    _SelfAdjointTransform.__name__ = _SelfAdjointTransform.__qualname__ = name
    _SelfAdjointTransform.__doc__ = doc
    return _SelfAdjointTransform


def make_fwht_module(function, doc):
    """``nn.Module`` wrapper calling ``function.apply`` (the reference's ``FWHT`` modules)."""

    class FWHT(nn.Module):
        def forward(self, x):
            return function.apply(x)

    FWHT.__doc__ = doc
    return FWHT
