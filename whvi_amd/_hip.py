"""ctypes binding of ``libwhvi_hip.so`` (C ABI: ``include/whvi_hip.h``).

This is the only place the package touches the native library.  There is NO fallback: if the
library is missing or a call fails, a ``RuntimeError`` is raised -- a CUDA/HIP tensor is never
silently routed to a CPU implementation.

torch is used here for plumbing only: ``data_ptr()``, the current HIP stream and the device
guard.  The library links ``libamdhip64.so.7``; torch is imported first so that the HIP runtime
torch already loaded (same SONAME) is the one the kernels are registered with and the one whose
streams we launch on.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  (must be loaded before the HIP library, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# The shipped loader reads NO environment: it loads the library built next to this file, or nothing.  (Measurement builds of
# the same library are loaded by the probes under tools/ by assigning LIB_PATH before the first call: tools/_tuning.py.)
LIB_PATH = os.path.join(_HERE, "libwhvi_hip.so")

AXIS_ROW, AXIS_COL = 0, 1
F32, F64, F16, I32, BF16 = 0, 1, 2, 3, 4

_DTYPE_CODE = {
    torch.float32: F32, torch.float64: F64, torch.float16: F16,
    torch.int32: I32, torch.bfloat16: BF16,
}
_DTYPE_SUFFIX = {
    torch.float32: "f32", torch.float64: "f64", torch.float16: "f16",
    torch.int32: "i32", torch.bfloat16: "bf16",
}

_lock = threading.Lock()
_lib = None


def _declare(lib):
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    lib.whvi_hip_abi_version.restype = ctypes.c_int
    lib.whvi_hip_abi_version.argtypes = []
    lib.whvi_last_error.restype = ctypes.c_char_p
    lib.whvi_last_error.argtypes = []
    lib.whvi_max_log2d.restype = ctypes.c_int
    lib.whvi_max_log2d.argtypes = [i32]
    lib.whvi_last_kernel.restype = ctypes.c_int
    lib.whvi_last_kernel.argtypes = [ctypes.c_char_p, i32]
    for sfx in ("f32", "f64", "f16", "bf16", "i32"):
        fn = getattr(lib, "whvi_fwht_" + sfx)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, vp, i64, i32, vp]
    lib.whvi_fwht_ex.restype = ctypes.c_int
    lib.whvi_fwht_ex.argtypes = [vp, vp, i64, i32, i32, i32, vp]
    for sfx in ("f32", "f64"):
        fn = getattr(lib, "whvi_fused_shs_" + sfx)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, vp, vp, vp, vp, i64, i32, i64, i64, i64, i32, vp]
        fn = getattr(lib, "whvi_fused_shs_ex_" + sfx)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, vp, vp, vp, vp, i64, i32, i64, i64, i64, i32, i32, vp]


def _declare_f3(lib):
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.whvi_reparam_kl_blocks.restype = ctypes.c_int
    lib.whvi_reparam_kl_blocks.argtypes = [i64]
    lib.whvi_reparam_kl_f32.restype = ctypes.c_int
    lib.whvi_reparam_kl_f32.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, ctypes.c_float, vp]
    lib.whvi_reparam_kl_philox_f32.restype = ctypes.c_int
    lib.whvi_reparam_kl_philox_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, ctypes.c_float, vp]
    lib.whvi_reparam_kl_bwd_f32.restype = ctypes.c_int
    lib.whvi_reparam_kl_bwd_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, ctypes.c_float, vp]
    p64 = ctypes.POINTER(ctypes.c_int64)
    lib.whvi_gauss_mnll_blocks.restype = ctypes.c_int
    lib.whvi_gauss_mnll_blocks.argtypes = [i64]
    lib.whvi_gauss_mnll_f32.restype = ctypes.c_int
    lib.whvi_gauss_mnll_f32.argtypes = [vp, vp, vp, vp, p64, p64, p64, ctypes.c_float, vp]
    lib.whvi_decay_lr_step.restype = ctypes.c_int
    lib.whvi_decay_lr_step.argtypes = [vp, vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int32, vp]
    lib.whvi_gauss_mnll_bwd_f32.restype = ctypes.c_int
    lib.whvi_gauss_mnll_bwd_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, p64, p64, p64, ctypes.c_float, vp]
    for sfx in ("f32", "f64"):
        fn = getattr(lib, "whvi_wbar_bwd_" + sfx)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, ctypes.c_int32, ctypes.c_int32, vp]
        fn = getattr(lib, "whvi_wbar_fwd_" + sfx)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, ctypes.c_int32, i64, i64, vp]
        fn = getattr(lib, "whvi_wbar_fwd_mean_" + sfx)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, vp, vp, vp, i64, i64, i64, ctypes.c_int32, vp]
        fn = getattr(lib, "whvi_diag_apply_" + sfx)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, ctypes.c_int32, ctypes.c_int32, vp]
        fn = getattr(lib, "whvi_diag_apply_bwd_" + sfx)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, ctypes.c_int32, i64, ctypes.c_int32, vp]
    lib.whvi_small_k_apply_f32.restype = ctypes.c_int
    lib.whvi_small_k_apply_f32.argtypes = [vp, vp, vp, vp, i64, i64, i64, ctypes.c_int32, ctypes.c_int32, vp]
    lib.whvi_row_dot_f32.restype = ctypes.c_int
    lib.whvi_row_dot_f32.argtypes = [vp, vp, vp, vp, i64, i64, ctypes.c_int32, ctypes.c_int32, vp]
    lib.whvi_stream_copy_probe.restype = ctypes.c_int
    lib.whvi_stream_copy_probe.argtypes = [vp, vp, i64, vp]
    lib.whvi_diag_apply_bwd_slabs.restype = ctypes.c_int64
    lib.whvi_diag_apply_bwd_slabs.argtypes = [ctypes.c_int32, i64, i64, ctypes.c_int32]


def lib():
    """Load (once) and return the ctypes handle; raise loudly when it is not there."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"whvi_amd: native HIP library not found at {LIB_PATH}. Build it with "
                        "`python -c 'import __graft_entry__ as g; g.build()'` or "
                        "`make -C whvi_amd/csrc -j8`. There is no CPU fallback for GPU tensors.")
                handle = ctypes.CDLL(LIB_PATH)
                _declare(handle)
                _declare_f3(handle)
                if handle.whvi_hip_abi_version() != 1:
                    raise RuntimeError("whvi_amd: libwhvi_hip.so ABI version mismatch")
                _lib = handle
    return _lib


def is_built() -> bool:
    return os.path.exists(LIB_PATH)


def last_error() -> str:
    return lib().whvi_last_error().decode()


def last_kernel() -> str:
    """Demangled symbol of the kernel instantiation this thread's last FWHT / fused launch selected ("" before any)."""
    buf = ctypes.create_string_buffer(256)
    lib().whvi_last_kernel(buf, 256)
    return buf.value.decode()


def max_log2d(dtype: torch.dtype) -> int:
    return lib().whvi_max_log2d(_DTYPE_CODE[dtype])


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor):
    """hipStream_t of torch's current stream on t's device (raw-handle fast path: no Stream object)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _OnDevice:
    """``with _OnDevice(dev):`` == ``with torch.cuda.device(dev):`` but free when dev is already current
    (the kernels launch on the calling thread's current device; the reference never sets it)."""
    __slots__ = ("dev", "ctx")

    def __init__(self, dev):
        self.dev, self.ctx = dev, None

    def __enter__(self):
        idx = self.dev.index
        if idx is not None and idx != torch.cuda.current_device():
            self.ctx = torch.cuda.device(self.dev)
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def _aligned(t: torch.Tensor) -> torch.Tensor:
    """Contiguous and 16-byte aligned view/copy of t (fresh torch allocations are 512-B aligned)."""
    t = t.contiguous()
    if t.data_ptr() % 16 != 0:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


FWHT_SIGNED_LANES = 1 << 23      # whvi_fwht_ex variant bit (include/whvi_hip.h): faster f32 / f64 streams, -0 results become +0


def fwht_rows(src: torch.Tensor, out: torch.Tensor = None, variant: int = None, signed_lanes: bool = False) -> torch.Tensor:
    """FWHT of every row of a contiguous (rows, D) device tensor.  ``out`` may alias ``src``
    (in place); by default a new tensor is returned and ``src`` is left untouched.  ``signed_lanes``: opt in to
    WHVI_FWHT_SIGNED_LANES (streams of f32 D = 512 .. 2048 / f64 D = 64 .. 2048: +1.7 %, the sign of a zero result is lost)."""
    if signed_lanes:
        variant = (variant or 0) | FWHT_SIGNED_LANES
    if src.device.type != "cuda":
        raise RuntimeError("X must be a CUDA tensor")
    if src.dim() != 2:
        raise RuntimeError("X must be two-dimensional")
    rows, d = src.shape
    if d < 1 or (d & (d - 1)) != 0:
        raise RuntimeError("n must be a power of 2")
    if src.dtype not in _DTYPE_CODE:
        raise RuntimeError(f"fwht: unsupported dtype {src.dtype} (float32/float64/float16/bfloat16/int32)")
    log2d = d.bit_length() - 1
    if out is None:
        src = _aligned(src)
        out = torch.empty_like(src, memory_format=torch.contiguous_format)
    else:
        if not (src.is_contiguous() and out.is_contiguous()):
            raise RuntimeError("fwht: src and out must be contiguous when out= is given")
        if out.shape != src.shape or out.dtype != src.dtype or out.device != src.device:
            raise RuntimeError("fwht: out must match src in shape, dtype and device")
    L = lib()
    with _OnDevice(src.device):
        if variant is None:
            rc = getattr(L, "whvi_fwht_" + _DTYPE_SUFFIX[src.dtype])(
                out.data_ptr(), src.data_ptr(), rows, log2d, _stream(src))
        else:
            rc = L.whvi_fwht_ex(out.data_ptr(), src.data_ptr(), rows, log2d,
                                _DTYPE_CODE[src.dtype], int(variant), _stream(src))
    _check(rc, "whvi_fwht")
    return out


def stream_copy_probe(src: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """whvi_stream_copy_probe: copy ``src`` (contiguous, a multiple of 16 bytes) with the transform kernels' streaming
    geometry and no arithmetic; ``out`` may be ``src`` (in place).  A measurement aid (bench.py: roofline.ceiling_measured)."""
    if src.device.type != "cuda" or not src.is_contiguous():
        raise RuntimeError("stream_copy_probe: contiguous CUDA tensor expected")
    if out is None:
        out = torch.empty_like(src)
    elif not out.is_contiguous() or out.numel() * out.element_size() != src.numel() * src.element_size() or out.device != src.device:
        raise RuntimeError("stream_copy_probe: out must match src")
    with _OnDevice(src.device):
        rc = lib().whvi_stream_copy_probe(out.data_ptr(), src.data_ptr(), src.numel() * src.element_size(), _stream(src))
    _check(rc, "whvi_stream_copy_probe")
    return out


FUSED_A_PER_SAMPLE, FUSED_C_PER_SAMPLE, FUSED_SRC_SHARED, FUSED_ONE_TRANSFORM = 1, 2, 4, 8


def fused_src_shared_supported(dtype: torch.dtype, d: int) -> bool:
    """Row lengths the shared-source form of ``whvi_fused_shs_*`` covers (rows of at least 64 sixteen-byte chunks)."""
    return fused_supported(dtype, d) and d * (4 if dtype == torch.float32 else 8) >= 1024


def fused_supported(dtype: torch.dtype, d: int) -> bool:
    """Row lengths ``whvi_fused_shs_*`` covers: one wavefront tile, D <= 8192 (f32) / 4096 (f64).  Longer rows have the
    plain transform only (``fwht_rows``: a block per row, then passes)."""
    if dtype == torch.float32:
        return 1 <= d <= 8192
    return dtype == torch.float64 and 1 <= d <= 4096


def fused_shs(src, a=None, b=None, c=None, *, axis: str = "col", n_samples: int = 1,
              sample_stride: int = 1, group_rows: int = 1, rows: int = None, d: int = None,
              dtype=None, device=None, out: torch.Tensor = None, a_per_sample: bool = False,
              c_per_sample: bool = False, src_shared: bool = False, one_transform: bool = False) -> torch.Tensor:
    """out[r] = a (.) FWHT(b_s (.) FWHT(c (.) src[r])) in ONE kernel (include/whvi_hip.h).

    ``src=None`` (axis="row", group_rows == d) synthesises the identity matrix per group, so
    with ``c = s2`` the input is ``torch.diag(s2)`` of src/weights.py:73 without reading HBM.
    ``src_shared`` (axis="col"): ``src`` is ``(sample_stride, d)`` -- ONE sample's rows, shared by all ``n_samples``
    samples -- and the result has ``n_samples * sample_stride`` rows in (sample, row) order (WHVI_FUSED_SRC_SHARED).
    ``one_transform`` (axis="col", ``c`` must be None): ``out[r] = a (.) FWHT(b_s (.) src[r])``, the second half alone.
    """
    ax = {"row": AXIS_ROW, "col": AXIS_COL}[axis]
    if src is not None:
        if src.device.type != "cuda" or src.dim() != 2:
            raise RuntimeError("fused_shs: src must be a 2-D CUDA tensor")
        src = _aligned(src)
        rows, d = src.shape
        dtype, device = src.dtype, src.device
        if src_shared:
            if ax != AXIS_COL or rows != sample_stride or not fused_src_shared_supported(dtype, d):
                raise RuntimeError("fused_shs: src_shared needs axis='col', src of sample_stride rows and rows of >= 1 KiB")
            rows = n_samples * sample_stride
    if dtype not in (torch.float32, torch.float64):
        raise RuntimeError("fused_shs: float32 / float64 only")
    if d < 1 or (d & (d - 1)) != 0:
        raise RuntimeError("n must be a power of 2")
    log2d = d.bit_length() - 1

    def prep(v, n):
        if v is None:
            return None
        v = _aligned(v.to(device=device, dtype=dtype).reshape(-1))
        if v.numel() != n:
            raise RuntimeError(f"fused_shs: scale vector has {v.numel()} elements, expected {n}")
        return v

    unit = group_rows if ax == AXIS_ROW else d
    a_ = prep(a, unit * (n_samples if a_per_sample else 1))
    b_ = prep(b, unit * n_samples)
    c_ = prep(c, unit * (n_samples if c_per_sample else 1))
    if one_transform and (c is not None or ax != AXIS_COL or src is None or not fused_src_shared_supported(dtype, d)):
        raise RuntimeError("fused_shs: one_transform needs axis='col', a source, c=None and rows of >= 1 KiB")
    flags = ((FUSED_A_PER_SAMPLE if a_per_sample else 0) | (FUSED_C_PER_SAMPLE if c_per_sample else 0) |
             (FUSED_SRC_SHARED if src_shared else 0) | (FUSED_ONE_TRANSFORM if one_transform else 0))
    if out is None:
        out = torch.empty((rows, d), dtype=dtype, device=device)
    elif not out.is_contiguous() or tuple(out.shape) != (rows, d) or out.dtype != dtype:
        raise RuntimeError("fused_shs: bad out tensor")
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    fn = getattr(lib(), "whvi_fused_shs_ex_" + _DTYPE_SUFFIX[dtype])
    with _OnDevice(device):
        rc = fn(out.data_ptr(), ptr(src), ptr(a_), ptr(b_), ptr(c_), rows, log2d, n_samples,
                sample_stride, group_rows, ax, flags, _stream(out))
    _check(rc, "whvi_fused_shs")
    return out


def reparam_kl(g_mu: torch.Tensor, g_rho: torch.Tensor, eps: torch.Tensor, lambda_: float):
    """One launch: (u (J, 1+S, D), sigma (J, D), kl (J,)) from g_mu, g_rho (J, D) and eps (J, S, D);
    see whvi_reparam_kl_f32 in include/whvi_hip.h."""
    if g_mu.device.type != "cuda" or g_mu.dtype != torch.float32:
        raise RuntimeError("reparam_kl: float32 CUDA tensors only")
    J, D = g_mu.shape
    S = eps.shape[1]
    g_mu, g_rho, eps = g_mu.contiguous(), g_rho.contiguous(), eps.contiguous()
    L = lib()
    nblk = (D + 255) // 256          # == whvi_reparam_kl_blocks(D)
    u = torch.empty((J, S + 1, D), dtype=torch.float32, device=g_mu.device)
    sigma = torch.empty((J, D), dtype=torch.float32, device=g_mu.device)
    part = torch.empty((J, nblk), dtype=torch.float32, device=g_mu.device)
    with _OnDevice(g_mu.device):
        rc = L.whvi_reparam_kl_f32(u.data_ptr(), sigma.data_ptr(), part.data_ptr(), g_mu.data_ptr(), g_rho.data_ptr(),
                                   eps.data_ptr() if S > 0 else None, J, S, D, float(lambda_), _stream(g_mu))
    _check(rc, "whvi_reparam_kl")
    return u, sigma, (part.sum(dim=1) if nblk > 1 else part[:, 0])


def new_rng_state(device, seed: int = None) -> torch.Tensor:
    """Device-resident generator state for ``reparam_kl_philox``: int64 [seed, launch offset, scratch].  Without an
    explicit seed one is drawn from torch's default CPU generator, so ``torch.manual_seed`` makes runs repeatable."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return torch.tensor([int(seed), 0, 0], dtype=torch.int64, device=device)


def reparam_kl_philox(g_mu: torch.Tensor, g_rho: torch.Tensor, n_samples: int, lambda_: float, state: torch.Tensor):
    """One launch, eps drawn in the kernel: (u (J, 1+S, D), sigma (J, D), kl (J,), eps (J, S, D)); see
    whvi_reparam_kl_philox_f32 in include/whvi_hip.h.  ``state`` comes from ``new_rng_state`` and is advanced in place."""
    if g_mu.device.type != "cuda" or g_mu.dtype != torch.float32:
        raise RuntimeError("reparam_kl_philox: float32 CUDA tensors only")
    if state.dtype != torch.int64 or state.numel() < 3 or state.device != g_mu.device or not state.is_contiguous():
        raise RuntimeError("reparam_kl_philox: state must be a contiguous int64 tensor of 3 words on the same device")
    J, D = g_mu.shape
    S = int(n_samples)
    g_mu, g_rho = g_mu.contiguous(), g_rho.contiguous()
    L = lib()
    nblk = (D + 255) // 256
    dev = g_mu.device
    u = torch.empty((J, S + 1, D), dtype=torch.float32, device=dev)
    sigma = torch.empty((J, D), dtype=torch.float32, device=dev)
    part = torch.empty((J, nblk), dtype=torch.float32, device=dev)
    eps = torch.empty((J, S, D), dtype=torch.float32, device=dev)
    with _OnDevice(dev):
        rc = L.whvi_reparam_kl_philox_f32(u.data_ptr(), sigma.data_ptr(), part.data_ptr(), eps.data_ptr() if S > 0 else None,
                                          g_mu.data_ptr(), g_rho.data_ptr(), state.data_ptr(), J, S, D, float(lambda_),
                                          _stream(g_mu))
    _check(rc, "whvi_reparam_kl_philox")
    return u, sigma, (part.sum(dim=1) if nblk > 1 else part[:, 0]), eps


def wbar_bwd_supported(dtype: torch.dtype, d: int) -> bool:
    """Shapes the one-launch backward covers (rows of one 16-byte chunk up to one wavefront tile)."""
    if dtype == torch.float32:
        return 4 <= d <= 8192
    return dtype == torch.float64 and 2 <= d <= 4096


def wbar_fwd(s1: torch.Tensor, u: torch.Tensor, s2: torch.Tensor, rows: int = None, base: torch.Tensor = None,
             first: int = 0, count: int = None):
    """One launch: W (J, S, R, D) with W[j,k] = the first R rows of S1_j fwht(diag(u[j, first + k]) fwht(diag(s2_j))),
    k < S = ``count`` (default: all rows of ``u`` from ``first`` on), plus ``base`` (J, R, D) when given; see
    whvi_wbar_fwd_f32 in include/whvi_hip.h."""
    if u.device.type != "cuda" or u.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("wbar_fwd: float32 / float64 CUDA tensors only")
    J, G, D = u.shape
    S = G - first if count is None else int(count)
    R = D if rows is None else int(rows)
    if first < 0 or S < 0 or first + S > G:
        raise RuntimeError("wbar_fwd: rows first .. first + count do not fit u")
    if tuple(s1.shape) != (J, D) or tuple(s2.shape) != (J, D) or (base is not None and tuple(base.shape) != (J, R, D)):
        raise RuntimeError("wbar_fwd: operand shapes do not match u")
    if not (s1.dtype == s2.dtype == u.dtype) or (base is not None and base.dtype != u.dtype):
        raise RuntimeError("wbar_fwd: operand dtypes do not match u")
    s1, u, s2 = s1.contiguous(), u.contiguous(), s2.contiguous()
    base = None if base is None else base.contiguous()
    out = torch.empty((J, S, R, D), dtype=u.dtype, device=u.device)
    fn = getattr(lib(), "whvi_wbar_fwd_" + _DTYPE_SUFFIX[u.dtype])
    with _OnDevice(u.device):
        rc = fn(out.data_ptr(), s1.data_ptr(), u.data_ptr(), s2.data_ptr(), None if base is None else base.data_ptr(),
                J, S, R, D.bit_length() - 1, G, int(first), _stream(u))
    _check(rc, "whvi_wbar_fwd")
    return out


# results up to this size take the one-launch mean + sample form (whvi_wbar_fwd_mean); beyond it the two-launch form wins
# (half the arithmetic per byte written once the write stream itself is the bound).  Measured, interleaved, HIP events
# (tools/probe_wbar_mean.py, profiles/r03/wbar_mean_one_vs_two_launches.log): 256 matrices of D = 4 x 16 samples 7.5 vs
# 16.3 us, D = 512 x 32 (32 MiB, config 2) 12.9 vs 16.8 us; D = 1024 x 16 (64 MiB) 21.1 vs 17.1, 128 MiB 35 vs 25
WBAR_INLINE_MEAN_MAX_BYTES = 48 << 20


def wbar_fwd_mean(s1: torch.Tensor, u: torch.Tensor, s2: torch.Tensor, rows: int = None, inline: bool = None):
    """W (J, S, R, D) with W[j,k] = w_bar(u[j,0]) + w_bar(u[j,1+k]) for u (J, 1 + S, D) -- src/weights.py:93.  One launch
    computing both terms (``whvi_wbar_fwd_mean``) for cache-resident results, otherwise the mean matrix once and every
    sample's matrix with the mean added in its epilogue (two launches of ``whvi_wbar_fwd``).  Same bits either way;
    ``inline`` forces the choice (tests, A/B)."""
    if u.device.type != "cuda" or u.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("wbar_fwd_mean: float32 / float64 CUDA tensors only")
    J, G, D = u.shape
    S, R = G - 1, (D if rows is None else int(rows))
    if S < 0 or tuple(s1.shape) != (J, D) or tuple(s2.shape) != (J, D) or not (s1.dtype == s2.dtype == u.dtype):
        raise RuntimeError("wbar_fwd_mean: operands do not match u")
    fits = D <= (4096 if u.dtype == torch.float32 else 2048)      # two 64-register tiles per wave
    if inline is None:
        inline = fits and J * S * R * D * u.element_size() <= WBAR_INLINE_MEAN_MAX_BYTES
    elif inline and not fits:
        raise RuntimeError("wbar_fwd_mean: rows this long have the two-launch form only")
    if not inline:
        mean = wbar_fwd(s1, u, s2, R, first=0, count=1)                        # (J, 1, R, D): w_bar(g_mu), once
        return wbar_fwd(s1, u, s2, R, base=mean.view(J, R, D), first=1)        # + w_bar(g_sigma * eps_k), per sample
    s1, u, s2 = s1.contiguous(), u.contiguous(), s2.contiguous()
    out = torch.empty((J, S, R, D), dtype=u.dtype, device=u.device)
    fn = getattr(lib(), "whvi_wbar_fwd_mean_" + _DTYPE_SUFFIX[u.dtype])
    with _OnDevice(u.device):
        rc = fn(out.data_ptr(), s1.data_ptr(), u.data_ptr(), s2.data_ptr(), J, S, R, D.bit_length() - 1, _stream(u))
    _check(rc, "whvi_wbar_fwd_mean")
    return out


def wbar_bwd(grad_w: torch.Tensor, s1: torch.Tensor, u: torch.Tensor, s2: torch.Tensor, mean: bool = False,
             no_lds: bool = False, tiles: str = None):
    """One launch: a (3, J, U, D) tensor [grad_u, part_s1, part_s2] from grad_w (J, S, R, D), s1 / s2 (J, D) and
    u (J, U, D), U = S -- or 1 + S with ``mean`` (W[j,k] = w_bar(u[j,0]) + w_bar(u[j,1+k]); slot 0 of the result is
    then left for the caller's sum over slots 1..S).  Entries i >= R are zero.  ``no_lds`` (tuning / cross-check):
    WHVI_WBAR_NO_LDS, the DPP butterfly network instead of the LDS-staged one; ``tiles`` = "small" / "big" forces or
    forbids the quarter-size tiles of small problems.  See whvi_wbar_bwd_f32."""
    if grad_w.device.type != "cuda" or grad_w.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("wbar_bwd: float32 / float64 CUDA tensors only")
    J, S, R, D = grad_w.shape
    U = S + 1 if mean else S
    if tuple(u.shape) != (J, U, D) or tuple(s1.shape) != (J, D) or tuple(s2.shape) != (J, D):
        raise RuntimeError("wbar_bwd: operand shapes do not match grad_w")
    if not (u.dtype == s1.dtype == s2.dtype == grad_w.dtype):
        raise RuntimeError("wbar_bwd: operand dtypes do not match grad_w")
    grad_w, s1, u, s2 = grad_w.contiguous(), s1.contiguous(), u.contiguous(), s2.contiguous()
    alloc = torch.empty if R == D else torch.zeros
    out = alloc((3, J, U, D), dtype=grad_w.dtype, device=grad_w.device)
    fn = getattr(lib(), "whvi_wbar_bwd_" + _DTYPE_SUFFIX[grad_w.dtype])
    with _OnDevice(grad_w.device):
        rc = fn(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), grad_w.data_ptr(), s1.data_ptr(),
                u.data_ptr(), s2.data_ptr(), J, S, R, D.bit_length() - 1, (1 if mean else 0) | (2 if no_lds else 0) | {None: 0, "small": 4, "big": 8}[tiles],
                _stream(grad_w))
    _check(rc, "whvi_wbar_bwd")
    return out


DIAG_X_SHARED, DIAG_MEAN_PLUS, DIAG_RELU_IN, DIAG_RELU_OUT = 1, 2, 4, 8
DIAG_TUNE_NT, DIAG_TUNE_CACHED, DIAG_TUNE_PLAIN_ORDER = 16, 32, 64      # tuning / cross-check flags (include/whvi_hip.h)


def diag_apply_supported(dtype: torch.dtype, d: int) -> bool:
    """Row lengths ``whvi_diag_apply_*`` covers: one 16-byte chunk up to one 64-register wavefront tile."""
    if dtype == torch.float32:
        return 4 <= d <= 4096 and (d & (d - 1)) == 0
    return dtype == torch.float64 and 2 <= d <= 2048 and (d & (d - 1)) == 0


def _diag_operands(x, s1, s2, u, n_samples, mean_plus, what, relu_in=False, relu_out=False):
    if x.device.type != "cuda" or x.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"{what}: float32 / float64 CUDA tensors only")
    D = x.shape[-1]
    if not diag_apply_supported(x.dtype, D):
        raise RuntimeError(f"{what}: D = {D} is outside the supported range for {x.dtype}")
    S = int(n_samples)
    if x.dim() == 3 and x.shape[0] == S:
        shared, B = False, x.shape[1]
    elif x.dim() == 2:
        shared, B = True, x.shape[0]
    else:
        raise RuntimeError(f"{what}: x must be (n_samples, batch, D) or (batch, D)")
    U = S + (1 if mean_plus else 0)
    if tuple(u.shape) != (U, D) or tuple(s1.shape) != (D,) or tuple(s2.shape) != (D,):
        raise RuntimeError(f"{what}: operand shapes do not match x (u must be ({U}, {D}))")
    if not (u.dtype == s1.dtype == s2.dtype == x.dtype):
        raise RuntimeError(f"{what}: operand dtypes do not match x")
    flags = ((DIAG_X_SHARED if shared else 0) | (DIAG_MEAN_PLUS if mean_plus else 0) | (DIAG_RELU_IN if relu_in else 0) |
             (DIAG_RELU_OUT if relu_out else 0))
    return S, B, D, shared, flags


def diag_apply(x: torch.Tensor, s1: torch.Tensor, s2: torch.Tensor, u: torch.Tensor, bias: torch.Tensor = None, *,
               n_samples: int, mean_plus: bool = True, out: torch.Tensor = None, tune: int = 0, relu_in: bool = False,
               relu_out: bool = False) -> torch.Tensor:
    """One launch: ``out[k] = x[(k)] * (wd(u[0]) + wd(u[1 + k])) + bias`` -- ``h @ (w_bar(g_mu) + w_bar(g_sigma eps_k)).T``
    of src/weights.py:87-93 for all MC samples without the matrices; see whvi_diag_apply_f32 in include/whvi_hip.h.
    ``x``: (S, B, D) or a shared (B, D); ``u``: (1 + S, D), or (S, D) with ``mean_plus=False``; returns (S, B, D).
    ``relu_in`` / ``relu_out``: an ``nn.ReLU`` in front of / behind the layer fused into the launch (WHVI_DIAG_RELU_*)."""
    S, B, D, shared, flags = _diag_operands(x, s1, s2, u, n_samples, mean_plus, "diag_apply", relu_in, relu_out)
    x, s1, s2, u = _aligned(x), _aligned(s1), _aligned(s2), _aligned(u)
    if bias is not None:
        if bias.numel() != D or bias.dtype != x.dtype:
            raise RuntimeError("diag_apply: bias must hold D elements of x's dtype")
        bias = _aligned(bias.reshape(-1))
    if out is None:
        out = torch.empty((S, B, D), dtype=x.dtype, device=x.device)
    elif not out.is_contiguous() or tuple(out.shape) != (S, B, D) or out.dtype != x.dtype or out.data_ptr() % 16:
        raise RuntimeError("diag_apply: bad out tensor")
    fn = getattr(lib(), "whvi_diag_apply_" + _DTYPE_SUFFIX[x.dtype])
    with _OnDevice(x.device):
        rc = fn(out.data_ptr(), x.data_ptr(), s1.data_ptr(), s2.data_ptr(), u.data_ptr(),
                None if bias is None else bias.data_ptr(), S, B, D.bit_length() - 1, flags | int(tune), _stream(x))
    _check(rc, "whvi_diag_apply")
    return out


def diag_apply_bwd(grad_out: torch.Tensor, x: torch.Tensor, s1: torch.Tensor, s2: torch.Tensor, u: torch.Tensor, *,
                   n_samples: int, mean_plus: bool = True, need_grad_x: bool = True, tune: int = 0, bias: torch.Tensor = None,
                   relu_in: bool = False, relu_out: bool = False):
    """Backward of ``diag_apply`` in one call: ``(grad_x (S, B, D) or None, out (4, U, D))`` with the rows of ``out`` as
    whvi_diag_apply_bwd_f32 documents them (slot 0 dL/du, 1 / 2 the per-sample shares of dL/ds1 / dL/ds2, 3 of dL/dbias;
    with ``mean_plus`` row 0 is left for the caller's sum over rows 1 ..)."""
    S, B, D, shared, flags = _diag_operands(x, s1, s2, u, n_samples, mean_plus, "diag_apply_bwd", relu_in, relu_out)
    if bias is not None:
        bias = _aligned(bias.reshape(-1))
    if tuple(grad_out.shape) != (S, B, D) or grad_out.dtype != x.dtype:
        raise RuntimeError("diag_apply_bwd: grad_out must be (n_samples, batch, D) of x's dtype")
    grad_out, x, s1, s2, u = _aligned(grad_out), _aligned(x), _aligned(s1), _aligned(s2), _aligned(u)
    L = lib()
    log2d = D.bit_length() - 1
    U = u.shape[0]
    out = torch.empty((4, U, D), dtype=x.dtype, device=x.device)
    grad_x = torch.empty((S, B, D), dtype=x.dtype, device=x.device) if need_grad_x else None
    if S == 0 or B == 0:
        out.zero_()
        return grad_x, out
    n_slabs = int(L.whvi_diag_apply_bwd_slabs(_DTYPE_CODE[x.dtype], S, B, log2d))
    part = torch.empty((S, n_slabs, 2, D), dtype=x.dtype, device=x.device)
    fn = getattr(L, "whvi_diag_apply_bwd_" + _DTYPE_SUFFIX[x.dtype])
    with _OnDevice(x.device):
        rc = fn(None if grad_x is None else grad_x.data_ptr(), out.data_ptr(), part.data_ptr(), grad_out.data_ptr(),
                x.data_ptr(), s1.data_ptr(), s2.data_ptr(), u.data_ptr(), None if bias is None else bias.data_ptr(), S, B, log2d,
                n_slabs, flags | int(tune), _stream(x))
    _check(rc, "whvi_diag_apply_bwd")
    return grad_x, out


APPLY_RELU_IN, APPLY_RELU_OUT = 1, 2


def small_k_apply_supported(x: torch.Tensor, n_out: int) -> bool:
    """Shapes ``whvi_small_k_apply_f32`` covers: a float32 (B, 4) or (B, 8) GPU input, N a multiple of 4 that fits the LDS."""
    k = x.shape[-1]
    if not (x.device.type == "cuda" and x.dtype == torch.float32 and x.dim() == 2 and k in (4, 8) and n_out >= 4 and n_out % 4 == 0):
        return False
    cpr, tpr = n_out // 4, 1
    while tpr < 256 and cpr % (tpr * 2) == 0:
        tpr *= 2
    return cpr // tpr <= 4


def small_k_apply(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor = None, relu_out: bool = False) -> torch.Tensor:
    """One launch: ``out[s] = x @ w[s].T (+ bias)`` for a narrow input shared by all samples -- x (B, K), K in {4, 8};
    w (S, N, K); out (S, B, N); see whvi_small_k_apply_f32 in include/whvi_hip.h."""
    S, N, K = w.shape
    if not small_k_apply_supported(x, N) or w.dtype != torch.float32 or w.device != x.device or x.shape[1] != K:
        raise RuntimeError("small_k_apply: unsupported operands")
    x, w = _aligned(x), _aligned(w)
    if bias is not None:
        bias = _aligned(bias.reshape(-1))
        if bias.numel() != N:
            raise RuntimeError("small_k_apply: bias must hold N elements")
    B = x.shape[0]
    out = torch.empty((S, B, N), dtype=torch.float32, device=x.device)
    with _OnDevice(x.device):
        rc = lib().whvi_small_k_apply_f32(out.data_ptr(), x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                                          S, B, N, K.bit_length() - 1, APPLY_RELU_OUT if relu_out else 0, _stream(x))
    _check(rc, "whvi_small_k_apply")
    return out


def row_dot_supported(x: torch.Tensor) -> bool:
    d = x.shape[-1]
    return x.device.type == "cuda" and x.dtype == torch.float32 and x.dim() == 3 and 4 <= d <= 4096 and (d & (d - 1)) == 0


def row_dot(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor = None, relu_in: bool = False) -> torch.Tensor:
    """One launch: ``y[s, b] = x[s, b, :] . w[s] (+ bias)`` -- x (S, B, D), w (S, D), y (S, B, 1); see whvi_row_dot_f32."""
    if not row_dot_supported(x) or tuple(w.shape) != (x.shape[0], x.shape[2]) or w.dtype != torch.float32:
        raise RuntimeError("row_dot: unsupported operands")
    x, w = _aligned(x), _aligned(w)
    S, B, D = x.shape
    y = torch.empty((S, B, 1), dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = bias.reshape(-1).contiguous()
    with _OnDevice(x.device):
        rc = lib().whvi_row_dot_f32(y.data_ptr(), x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(), S, B,
                                    D.bit_length() - 1, APPLY_RELU_IN if relu_in else 0, _stream(x))
    _check(rc, "whvi_row_dot")
    return y


def reparam_kl_bwd(grad_u, grad_kl, g_mu, g_rho, eps, sigma, lambda_: float):
    """One launch: (grad_mu, grad_rho), each (J, D); see whvi_reparam_kl_bwd_f32 in include/whvi_hip.h.
    ``grad_u`` (J, 1+S, D) and ``grad_kl`` (J,) may be None (= zero)."""
    J, D = g_mu.shape
    S = eps.shape[1]
    grad_u = None if grad_u is None else grad_u.contiguous()
    grad_kl = None if grad_kl is None else grad_kl.contiguous()
    out = torch.empty((2, J, D), dtype=torch.float32, device=g_mu.device)
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with _OnDevice(g_mu.device):
        rc = lib().whvi_reparam_kl_bwd_f32(out[0].data_ptr(), out[1].data_ptr(), ptr(grad_u), ptr(grad_kl),
                                           g_mu.data_ptr(), g_rho.data_ptr(), eps.data_ptr() if S > 0 else None,
                                           sigma.data_ptr(), J, S, D, float(lambda_), _stream(g_mu))
    _check(rc, "whvi_reparam_kl_bwd")
    return out[0], out[1]


def _mnll_dims(y: torch.Tensor, y_hat: torch.Tensor):
    """Three (size, y_hat stride, y stride) triples, dimensions ordered by decreasing y_hat stride (memory order)."""
    order = sorted(range(3), key=lambda k: -y_hat.stride(k))
    arr = ctypes.c_int64 * 3
    return (arr(*[y_hat.size(k) for k in order]), arr(*[y_hat.stride(k) for k in order]),
            arr(*[y.stride(k) for k in order]))


def gauss_mnll(y: torch.Tensor, y_hat: torch.Tensor, sigma: torch.Tensor, scale: float):
    """Per-block partial sums (blocks, 2) of whvi_gauss_mnll_f32: ``y_hat`` (m, n_out, n_mc) with any strides,
    ``y`` an expanded view of the same shape, ``sigma`` a device scalar."""
    L = lib()
    part = torch.empty((L.whvi_gauss_mnll_blocks(y_hat.numel()), 2), dtype=torch.float32, device=y_hat.device)
    size, hs, ys = _mnll_dims(y, y_hat)
    with _OnDevice(y_hat.device):
        rc = L.whvi_gauss_mnll_f32(part.data_ptr(), y.data_ptr(), y_hat.data_ptr(), sigma.data_ptr(), size, hs, ys,
                                   float(scale), _stream(y_hat))
    _check(rc, "whvi_gauss_mnll")
    return part


def decay_lr_step(t: torch.Tensor, lr: torch.Tensor, base_lr: float, lambda0: float, gamma: float, p: float,
                  advance: bool = True) -> None:
    """whvi_decay_lr_step: ``t`` (0-d float64) += 1 when ``advance``; ``lr`` (0-d float32) <- base_lr * lambda0 *
    (1 + gamma t)^-p.  One single-thread launch on the current stream (capture-safe)."""
    if t.dtype != torch.float64 or lr.dtype != torch.float32 or t.numel() != 1 or lr.numel() != 1 or t.device != lr.device \
            or t.device.type != "cuda":
        raise RuntimeError("decay_lr_step: t must be a float64 and lr a float32 device scalar on the same GPU")
    with _OnDevice(t.device):
        rc = lib().whvi_decay_lr_step(t.data_ptr(), lr.data_ptr(), float(base_lr), float(lambda0), float(gamma), float(p),
                                      1 if advance else 0, _stream(t))
    _check(rc, "whvi_decay_lr_step")


def gauss_mnll_bwd(grad_out, part, y, y_hat, sigma, scale: float):
    """(grad_yhat with y_hat's strides, grad_sigma scalar) of whvi_gauss_mnll_bwd_f32."""
    grad_yhat = torch.empty_strided(y_hat.size(), y_hat.stride(), dtype=torch.float32, device=y_hat.device)
    grad_sigma = torch.empty((), dtype=torch.float32, device=y_hat.device)
    size, hs, ys = _mnll_dims(y, y_hat)
    with _OnDevice(y_hat.device):
        rc = lib().whvi_gauss_mnll_bwd_f32(grad_yhat.data_ptr(), grad_sigma.data_ptr(), grad_out.data_ptr(),
                                           part.data_ptr(), y.data_ptr(), y_hat.data_ptr(), sigma.data_ptr(), size, hs,
                                           ys, float(scale), _stream(y_hat))
    _check(rc, "whvi_gauss_mnll_bwd")
    return grad_yhat, grad_sigma
