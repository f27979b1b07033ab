"""The C-ABI library loads and exports every symbol include/whvi_hip.h declares; argument checks
that happen before any launch work without a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "whvi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(whvi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = _declared_symbols()
    for needed in ("whvi_fwht_f32", "whvi_fwht_f64", "whvi_fwht_f16", "whvi_fwht_bf16", "whvi_fwht_i32",
                   "whvi_fwht_ex", "whvi_fused_shs_f32", "whvi_fused_shs_f64", "whvi_last_error",
                   "whvi_hip_abi_version", "whvi_max_log2d"):
        assert needed in syms


def test_library_exports_every_declared_symbol():
    from whvi_amd import _hip
    assert _hip.is_built(), "libwhvi_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/whvi_hip.h but not exported"
    assert _hip.lib().whvi_hip_abi_version() == 1


def test_shipped_library_reads_no_environment():
    """A library a maintainer links must not change its launch form with the caller's environment: the shipped build
    (no -DWHVI_TUNING_BUILD, whvi_amd/csrc/tuning.hpp) contains no WHVI_* switch name, does not import getenv, and the
    product Makefile target never defines the macro.  Measurement builds live apart (make tuning -> whvi_amd/_exp/)."""
    import subprocess
    from whvi_amd import _hip
    default_lib = os.path.join(ROOT, "whvi_amd", "libwhvi_hip.so")
    assert os.path.exists(default_lib)
    strings = subprocess.run(["strings", "-a", default_lib], capture_output=True, text=True, check=True).stdout
    hits = sorted({ln.strip() for ln in strings.splitlines() if re.search(r"WHVI_[A-Z0-9_]+", ln)})
    assert hits == [], hits
    imports = subprocess.run(["nm", "-D", "--undefined-only", default_lib], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in imports
    mk = open(os.path.join(ROOT, "whvi_amd", "csrc", "Makefile")).read()
    product = mk.split("tuning:")[0]
    assert "WHVI_TUNING_BUILD" not in re.sub(r"#.*", "", product)
    # every run-time switch of the sources goes through the one macro that is a null pointer in the shipped build
    for name in os.listdir(os.path.join(ROOT, "whvi_amd", "csrc")):
        if name.endswith((".hpp", ".hip")) and name != "tuning.hpp":
            text = open(os.path.join(ROOT, "whvi_amd", "csrc", name)).read()
            assert "getenv" not in text, name


def test_argument_checks_without_gpu():
    from whvi_amd import _hip
    L = _hip.lib()
    buf = (ctypes.c_char * 256)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    assert L.whvi_fwht_f32(None, None, 4, 3, None) == -1 and "null" in _hip.last_error()
    assert L.whvi_fwht_f32(p16, p16, -1, 3, None) == -1
    assert L.whvi_fwht_f32(p16, p16, 1, 25, None) == -2 and "supported range" in _hip.last_error()
    assert L.whvi_fwht_f64(p16, p16, 1, 25, None) == -2
    assert L.whvi_fused_shs_f32(p16, p16, None, None, None, 1, 14, 1, 1, 1, 1, None) == -2   # fused: one-wave rows only
    assert L.whvi_fwht_f32(p16 + 4, p16 + 4, 1, 3, None) == -3 and "aligned" in _hip.last_error()
    assert L.whvi_fwht_f32(p16, p16 + 16, 1, 3, None) == -5          # partial overlap
    assert L.whvi_fwht_ex(p16, p16, 1, 3, 99, 0, None) == -1         # unknown dtype
    assert L.whvi_fwht_f32(None, None, 0, 3, None) == 0 and _hip.last_error() == ""   # empty batch
    assert L.whvi_fused_shs_f32(p16, p16, None, None, None, 1, 3, 0, 1, 1, 1, None) == -1  # n_samples < 1
    assert L.whvi_fused_shs_f32(p16, None, None, None, None, 1, 3, 1, 1, 9, 0, None) == -1  # identity needs group_rows <= D
    assert [L.whvi_max_log2d(i) for i in range(5)] == [24, 24, 16, 24, 16] and L.whvi_max_log2d(7) == -1


def test_argument_checks_of_the_fused_flags_without_gpu():
    """WHVI_FUSED_SRC_SHARED / WHVI_FUSED_ONE_TRANSFORM: the conditions include/whvi_hip.h states are checked before any
    launch -- an unknown flag, a shared source in place, a one-transform call with c, rows shorter than 1 KiB, the row
    axis -- and say which flag they are about."""
    from whvi_amd import _hip
    L = _hip.lib()
    buf = (ctypes.c_char * 65536)()
    p = (ctypes.addressof(buf) + 15) & ~15
    q = p + 32768
    fused = L.whvi_fused_shs_ex_f32
    assert fused(q, p, None, None, None, 4, 9, 2, 2, 1, 1, 16, None) == -1 and "unknown fused flags" in _hip.last_error()
    assert fused(p, p, None, None, None, 4, 9, 2, 2, 1, 1, 4, None) == -5 and "shared source" in _hip.last_error()   # in place
    assert fused(q, p, None, None, None, 4, 9, 2, 2, 1, 0, 4, None) == -1                                            # row axis
    assert fused(q, p, None, None, None, 4, 7, 2, 2, 1, 1, 4, None) == -1                                            # 512-byte rows
    assert fused(q, p, None, None, p, 4, 9, 2, 2, 1, 1, 8, None) == -1 and "one-transform" in _hip.last_error()      # c given
    assert fused(q, None, None, None, None, 4, 9, 2, 2, 1, 1, 8, None) == -1                                         # no source
    assert L.whvi_fused_shs_ex_f64(q, p, None, None, None, 4, 6, 2, 2, 1, 1, 8, None) == -1                          # f64: 512-byte rows
    assert fused(q, p, None, None, None, 0, 9, 2, 2, 1, 1, 12, None) == 0                                            # empty batch


def test_argument_checks_of_the_training_step_entry_points():
    """whvi_wbar_fwd / _bwd, whvi_reparam_kl_bwd, whvi_gauss_mnll: bad arguments are reported before any launch."""
    from whvi_amd import _hip
    L = _hip.lib()
    buf = (ctypes.c_char * 512)()
    p16 = (ctypes.addressof(buf) + 15) & ~15
    i64x3 = ctypes.c_int64 * 3
    assert L.whvi_wbar_fwd_f32(p16, p16, p16, p16, None, 1, 1, 4, 1, 1, 0, None) == -2 and "supported range" in _hip.last_error()
    assert L.whvi_wbar_fwd_f32(p16, p16, p16, p16, None, 1, 1, 9, 3, 1, 0, None) == -1 and "exceeds D" in _hip.last_error()
    assert L.whvi_wbar_fwd_f64(None, p16, p16, p16, None, 1, 1, 2, 1, 1, 0, None) == -1 and "null" in _hip.last_error()
    assert L.whvi_wbar_fwd_f32(p16 + 4, p16, p16, p16, None, 1, 1, 4, 2, 1, 0, None) == -3
    assert L.whvi_wbar_fwd_f32(p16, p16, p16, p16, None, 1, 2, 4, 2, 2, 1, None) == -1 and "u_group" in _hip.last_error()
    assert L.whvi_wbar_fwd_f32(None, None, None, None, None, 0, 3, 4, 2, 3, 0, None) == 0            # no matrices
    assert L.whvi_wbar_bwd_f32(p16, p16, p16, p16, p16, p16, p16, 1, 1, 4, 14, 0, None) == -2
    assert L.whvi_wbar_bwd_f32(p16, p16, p16, p16, p16, p16, p16, 1, 1, 4, 2, 16, None) == -1 and "flags" in _hip.last_error()
    assert L.whvi_wbar_bwd_f64(p16, p16, p16, None, p16, p16, p16, 1, 1, 2, 1, 0, None) == -1
    assert L.whvi_wbar_bwd_f32(None, None, None, None, None, None, None, 2, 0, 4, 2, 1, None) == 0
    assert L.whvi_reparam_kl_bwd_f32(p16, p16, None, None, p16, p16, p16, p16, 1, 1, 4, 0.0, None) == -1   # lambda <= 0
    assert L.whvi_reparam_kl_bwd_f32(None, p16, None, None, p16, p16, p16, p16, 1, 1, 4, 1.0, None) == -1
    assert L.whvi_reparam_kl_bwd_f32(None, None, None, None, None, None, None, None, 0, 1, 4, 1.0, None) == 0
    assert L.whvi_reparam_kl_philox_f32(p16, p16, p16, p16, p16, p16, None, 1, 1, 4, 1.0, None) == -1   # no generator state
    assert L.whvi_reparam_kl_philox_f32(p16, p16, p16, p16, p16, p16, p16 + 4, 1, 1, 4, 1.0, None) == -3
    assert L.whvi_reparam_kl_philox_f32(None, None, None, None, None, None, None, 0, 1, 4, 1.0, None) == 0
    assert L.whvi_gauss_mnll_f32(p16, p16, p16, p16, None, None, None, 1.0, None) == -1
    assert L.whvi_gauss_mnll_f32(p16, p16, p16, p16, i64x3(2, -1, 1), i64x3(1, 1, 1), i64x3(1, 1, 0), 1.0, None) == -1
    assert L.whvi_gauss_mnll_bwd_f32(None, None, p16, p16, p16, p16, p16, i64x3(1, 1, 1), i64x3(1, 1, 1), i64x3(1, 1, 0),
                                     1.0, None) == -1
    assert L.whvi_gauss_mnll_blocks(1) == 1 and L.whvi_gauss_mnll_blocks(10 ** 9) == 2048
    assert L.whvi_decay_lr_step(None, p16, 1.0, 1.0, 1.0, 1.0, 1, None) == -1 and "null" in _hip.last_error()
    assert L.whvi_decay_lr_step(p16 + 4, p16, 1.0, 1.0, 1.0, 1.0, 1, None) == -3
    assert L.whvi_decay_lr_step(p16, p16 + 2, 1.0, 1.0, 1.0, 1.0, 1, None) == -3


def test_last_kernel_before_any_launch():
    """whvi_last_kernel: nothing launched in this (GPU-less) process -> empty string, length 0; bad buffers are refused."""
    import ctypes
    from whvi_amd import _hip
    L = _hip.lib()
    buf = ctypes.create_string_buffer(8)
    assert L.whvi_last_kernel(buf, 8) == 0 and buf.value == b""
    assert L.whvi_last_kernel(None, 8) == -1 and L.whvi_last_kernel(buf, 0) == -1
    assert _hip.last_kernel() == ""


def test_gpu_tensors_never_fall_back(monkeypatch):
    """A missing native library is an error, not a CPU detour."""
    from whvi_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(RuntimeError, match="native HIP library not found"):
        _hip.lib()


def test_fwht_cuda_module_surface():
    import torch
    import fwht_cuda
    assert callable(fwht_cuda.fwht)
    with pytest.raises(RuntimeError, match="X must be a CUDA tensor"):   # fwht_cuda.cpp:6
        fwht_cuda.fwht(torch.randn(2, 4))
