"""Drop-in replacement for the reference's compiled extension module ``fwht_cpp``
(src/fwht/cpp/fwht.cpp:31-34: ``forward(x)``, ``backward(grad)``; imported at
src/fwht/cpp/fwht.py:4, test/walsh.py:5, benchmarks/walsh.py:4).

Same semantics as src/fwht/cpp/fwht.cpp:3-21: the transform runs along dimension 1 of a host
tensor of any rank >= 2 (2-D ``(batch, D)`` in the tests, ``(1, D, D)`` in benchmarks/walsh.py:21),
returns a new tensor, and is bit-identical to the reference (ascending butterfly strides).
Backed by the OpenMP library ``whvi_amd/libwhvi_cpu.so``.
"""
import torch

__all__ = ["forward", "backward"]


def _fwht_dim1(x: torch.Tensor) -> torch.Tensor:
    from whvi_amd import _cpu
    if x.dim() < 2:
        raise RuntimeError("fwht_cpp: expected a tensor with at least 2 dimensions")
    if x.dim() == 2:
        return _cpu.fwht_rows(x)
    moved = x.movedim(1, -1)
    shape = moved.shape
    out = _cpu.fwht_rows(moved.reshape(-1, shape[-1]))
    return out.reshape(shape).movedim(-1, 1)


def forward(x: torch.Tensor) -> torch.Tensor:
    return _fwht_dim1(x)


def backward(grad_output: torch.Tensor) -> torch.Tensor:
    # H is symmetric: the vector-Jacobian product is the same transform (fwht.cpp:27-29)
    return _fwht_dim1(grad_output)
