"""Drop-in replacement for the reference's compiled extension module ``fwht_cuda``.

The reference imports this name unconditionally (``import fwht_cuda`` at
src/fwht/cuda/fwht.py:2, hence src/weights.py:8, test/walsh.py:6, benchmarks/walsh_plot.py:11)
and calls exactly one function, ``fwht_cuda.fwht(X)`` (src/fwht/cuda/fwht_cuda.cpp:5-18).  This
module provides that function on top of the MI355X HIP library (``whvi_amd/libwhvi_hip.so``,
C ABI in ``include/whvi_hip.h``).  With the repo root on ``sys.path`` the reference's tests and
benchmarks run unchanged on the alias package ``src/``; to keep the reference's
OWN ``src`` package and swap only this module and ``fwht_cpp``, put ``dropin/`` on the path instead
(INTEGRATION.md 1).

Importing the module never touches the GPU or the native library (the reference's CPU path
imports it too); the library is loaded on the first call and a missing library is an error.
"""
import torch

__all__ = ["fwht"]


def fwht(X: torch.Tensor) -> torch.Tensor:
    """Batched fast Walsh-Hadamard transform of the rows of a 2-D GPU tensor.

    Same contract as src/fwht/cuda/fwht_cuda.cpp:5-14: ``X`` must be a CUDA (HIP) tensor,
    two-dimensional, last dimension a power of two -- violations raise ``RuntimeError`` with
    the reference's messages; a NEW tensor is returned and ``X`` is left untouched.
    Beyond the reference: float16 / bfloat16 / int32 inputs, D = 1, 2 and D up to 2^24 (16-bit types: 65536) work,
    launch errors are raised instead of ignored, and the kernel runs on torch's current stream
    of ``X``'s device.
    """
    from whvi_amd import _hip
    if not isinstance(X, torch.Tensor):
        raise TypeError("fwht(): argument 'X' must be a torch.Tensor")
    return _hip.fwht_rows(X)
