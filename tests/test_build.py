"""Build hygiene that needs no GPU: no shipped kernel may use scratch memory (a register spill
silently turns an HBM-bound kernel into a scratch-bound one)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_uses_scratch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_spills.py")],
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 with scratch" in out.stdout


def test_roofline_traffic_is_tied_to_the_build(monkeypatch):
    """bench.py reports the PMC-measured HBM traffic only for the kernel symbol AND the kernel sources it was collected
    on (profiles/hbm_traffic.json carries both); anything else yields null with the reason."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))["fwht_f32_D4096_rows1048576"]
    for field in ("hbm_bytes_per_launch", "kernel_symbol", "source_sha256", "algorithmic_bytes_per_launch"):
        assert field in rec
    assert 0.99 < rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"] < 1.05
    value, note = bench.recorded_traffic("fwht_f32_D4096_rows1048576", "whvi::some_other_kernel<float>")
    assert value is None and "this run launched" in note
    value, note = bench.recorded_traffic("fwht_f32_D512_rows7", rec["kernel_symbol"])
    assert value is None and "no PMC record" in note
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: rec["source_sha256"])
    value, note = bench.recorded_traffic("fwht_f32_D4096_rows1048576", rec["kernel_symbol"])
    assert value == rec["hbm_bytes_per_launch"]
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "0" * 64)
    value, note = bench.recorded_traffic("fwht_f32_D4096_rows1048576", rec["kernel_symbol"])
    assert value is None and "re-collect" in note


def test_setup_py_builds_an_installable_tree(tmp_path):
    """setup.py (the counterpart of the reference's src/fwht/{cuda,cpp}/setup.py): `build` runs make and lays out the
    package with both native libraries inside it and the two top-level modules the reference imports; the result works
    from a clean directory with nothing of the checkout on the path."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "setup.py", "-q", "build", "--build-base", str(tmp_path / "b")],
                         capture_output=True, text=True, timeout=1800, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lib = tmp_path / "b" / "lib"
    for rel in ("fwht_cuda.py", "fwht_cpp.py", "whvi_amd/libwhvi_hip.so", "whvi_amd/libwhvi_cpu.so", "whvi_amd/fwht/cuda.py"):
        assert (lib / rel).exists(), rel
    assert not (lib / "src").exists(), "the alias package must not be installed"
    code = ("import torch, fwht_cpp, fwht_cuda, whvi_amd, ctypes; from whvi_amd import _hip\n"
            "assert whvi_amd.__file__.startswith(%r), whvi_amd.__file__\n"
            "assert fwht_cpp.forward(torch.tensor([[1., 2, 3, 4]])).tolist() == [[10., -2., -4., 0.]]\n"
            "assert _hip.LIB_PATH.startswith(%r) and ctypes.CDLL(_hip.LIB_PATH).whvi_hip_abi_version() >= 1\n") % (str(lib), str(lib))
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(tmp_path),
                         env={**os.environ, "PYTHONPATH": str(lib)})
    assert run.returncode == 0, run.stderr[-2000:]
