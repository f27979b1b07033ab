"""ctypes binding of ``libwhvi_cpu.so`` -- the native FWHT for HOST tensors (the replacement for
the reference's ``fwht_cpp`` extension, src/fwht/cpp/fwht.cpp).  Never used for GPU tensors."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwhvi_cpu.so")
_lib = None

_SUFFIX = {torch.float32: "f32", torch.float64: "f64", torch.int32: "i32", torch.int64: "i64"}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"whvi_amd: {LIB_PATH} not built (make -C whvi_amd/csrc cpu)")
        handle = ctypes.CDLL(LIB_PATH)
        for sfx in _SUFFIX.values():
            fn = getattr(handle, "whvi_cpu_fwht_" + sfx)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        _lib = handle
    return _lib


def fwht_rows(x: torch.Tensor) -> torch.Tensor:
    """FWHT of every row of a 2-D host tensor; returns a new tensor."""
    if x.device.type != "cpu":
        raise RuntimeError("whvi_amd._cpu.fwht_rows handles host tensors only")
    if x.dtype in (torch.float16, torch.bfloat16):
        return fwht_rows(x.float()).to(x.dtype)   # f32 arithmetic, one rounding (as on the GPU)
    if x.dtype not in _SUFFIX:
        raise RuntimeError(f"fwht: unsupported dtype {x.dtype}")
    rows, n = x.shape
    if n < 1 or (n & (n - 1)) != 0:
        raise RuntimeError("n must be a power of 2")
    src = x.contiguous()
    out = torch.empty_like(src)
    rc = getattr(lib(), "whvi_cpu_fwht_" + _SUFFIX[x.dtype])(out.data_ptr(), src.data_ptr(), rows, n)
    if rc != 0:
        raise RuntimeError("whvi_cpu_fwht: bad arguments")
    return out
