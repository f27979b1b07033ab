"""ctypes/numpy binding of oracle/fwht_oracle.c (TEST INFRASTRUCTURE)."""
import ctypes
import os
import subprocess
import sys
import importlib

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_I64 = ctypes.c_int64
_P = ctypes.c_void_p


def build(force: bool = False) -> str:
    """Compile fwht_oracle.c with gcc (seconds).  Idempotent."""
    src = os.path.join(_HERE, "fwht_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "_build/liboracle.so"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        for suffix in ("f32", "f64", "i32", "i64"):
            fn = getattr(lib, f"oracle_fwht_{suffix}")
            fn.argtypes = [_P, _I64, _I64]
            fn.restype = None
        for suffix in ("f32", "f64"):
            fn = getattr(lib, f"oracle_fwht_desc_{suffix}")
            fn.argtypes = [_P, _I64, _I64]
            fn.restype = None
            fn = getattr(lib, f"oracle_pipeline_{suffix}")
            fn.argtypes = [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, ctypes.c_int]
            fn.restype = None
            fn = getattr(lib, f"oracle_pipeline_ex_{suffix}")
            fn.argtypes = [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, ctypes.c_int, ctypes.c_int]
            fn.restype = None
        lib.oracle_dense_wht_f64.argtypes = [_P, _P, _I64, _I64]
        lib.oracle_dense_wht_f64.restype = None
        _lib = lib
    return _lib


_SUFFIX = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64",
           np.dtype(np.int32): "i32", np.dtype(np.int64): "i64"}


def _is_pow2(n: int) -> bool:
    return n >= 1 and (n & (n - 1)) == 0


def fwht(x: np.ndarray) -> np.ndarray:
    """Unnormalised natural-order WHT of every row of a (rows, n) array, ascending strides
    (src/fwht/cpp/fwht.cpp:7-18).  float16 input follows the build's fp16 contract:
    fp32 arithmetic, one final rounding to fp16."""
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("oracle.fwht expects (rows, n)")
    if not _is_pow2(x.shape[1]):
        raise ValueError("n must be a power of 2")
    if x.dtype == np.float16:
        return fwht(x.astype(np.float32)).astype(np.float16)
    out = np.ascontiguousarray(x).copy()
    getattr(_load(), "oracle_fwht_" + _SUFFIX[out.dtype])(out.ctypes.data, out.shape[0], out.shape[1])
    return out


def fwht_descending(x: np.ndarray) -> np.ndarray:
    """Same network, strides n/2 .. 1 (the order of src/fwht/cuda/fwht_cuda_kernel.cu:94)."""
    x = np.asarray(x)
    out = np.ascontiguousarray(x).copy()
    getattr(_load(), "oracle_fwht_desc_" + _SUFFIX[out.dtype])(out.ctypes.data, out.shape[0], out.shape[1])
    return out


def pipeline(x, a=None, b=None, c=None, *, n_samples=1, sample_stride=1, group_rows=1, axis="col",
             a_per_sample=False, c_per_sample=False):
    """y = a (.) FWHT(b (.) FWHT(c (.) x)); see fwht_oracle.c:oracle_pipeline_f32.  ``a_per_sample`` /
    ``c_per_sample``: the outer vectors are indexed by the row's sample like ``b`` (oracle_pipeline_ex_<type>)."""
    x = np.ascontiguousarray(x)
    assert x.ndim == 2 and x.dtype in (np.float32, np.float64)
    rows, n = x.shape
    ax = {"row": 0, "col": 1}[axis]

    def prep(v, length):
        if v is None:
            return None, None
        v = np.ascontiguousarray(v, dtype=x.dtype).reshape(-1)
        assert v.size == length, (v.size, length)
        return v, v.ctypes.data

    unit = group_rows if ax == 0 else n
    a_, ap = prep(a, unit * (n_samples if a_per_sample else 1))
    b_, bp = prep(b, unit * n_samples)
    c_, cp = prep(c, unit * (n_samples if c_per_sample else 1))
    y = np.empty_like(x)
    if a_per_sample or c_per_sample:
        getattr(_load(), "oracle_pipeline_ex_" + _SUFFIX[x.dtype])(
            y.ctypes.data, x.ctypes.data, ap, bp, cp, rows, n, n_samples, sample_stride, group_rows, ax,
            (1 if a_per_sample else 0) | (2 if c_per_sample else 0))
    else:
        getattr(_load(), "oracle_pipeline_" + _SUFFIX[x.dtype])(
            y.ctypes.data, x.ctypes.data, ap, bp, cp, rows, n, n_samples, sample_stride, group_rows, ax)
    return y


def dense_wht(x: np.ndarray) -> np.ndarray:
    """(H @ x.T).T in float64 via the closed-form Hadamard entry (src/utils.py:88-101)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    _load().oracle_dense_wht_f64(y.ctypes.data, x.ctypes.data, x.shape[0], x.shape[1])
    return y


def hadamard(n: int) -> np.ndarray:
    """Dense Sylvester-Hadamard matrix, equal to src/utils.py:74-101 build_H(n)."""
    i = np.arange(n)
    bits = i[:, None] & i[None, :]
    par = np.zeros_like(bits)
    while bits.any():
        par ^= bits & 1
        bits >>= 1
    return (1 - 2 * par).astype(np.float64)


_ref_module = None


def load_reference_cpp():
    """Return the reference's own compiled C++ FWHT module (oracle/_ref), or None.

    Loaded by file path under a private handle: it is NOT put on sys.path / sys.modules, so
    ``import fwht_cpp`` elsewhere keeps resolving to the product's drop-in module."""
    global _ref_module
    if _ref_module is not None:
        return _ref_module
    ref_dir = os.path.join(_HERE, "_ref")
    if not os.path.isdir(ref_dir):
        return None
    paths = [os.path.join(ref_dir, f) for f in sorted(os.listdir(ref_dir)) if f.startswith("fwht_cpp") and f.endswith(".so")]
    if not paths:
        return None
    import importlib.util
    import torch  # noqa: F401  (the extension links libtorch)
    try:
        spec = importlib.util.spec_from_file_location("fwht_cpp", paths[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    except (ImportError, OSError):
        return None
    _ref_module = mod
    return mod
