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


# ---- what the SHIPPED binary contains (tools/shipped_isa.py: metadata + disassembly of libwhvi_hip.so's gfx950 code objects)
# kernel (as rocprofv3 prints it; the instantiations profiles/hbm_traffic.json and DESIGN.md's tables publish numbers for),
# VGPR budget of the occupancy it was tuned at (waves per SIMD = 512 // VGPRs: 128 -> 4, 168 -> 3), store mnemonic of its
# streaming path, and how those stores must be issued: "spaced" = at most 6 of them in back-to-back groups (the read + write streams: the
# headline runs 6.44 TB/s with one instruction between its 16 stores and 5.89 with them back to back, kernels.hpp /
# profiles/r03/rows_store_issue_ab.log), "adjacent" = runs of >= 3 (the write-only streams, where spacing LOSES 6-9 %:
# wbar_fwd.hpp, profiles/r03/write_stream_store_form_ab.log)
ISSUE_CONTRACTS = [
    ("whvi::fwht_rows_kernel<float, 12, 16, 0, false, true, 256, 1, false>", 128, "buffer_store_dwordx4", "spaced"),    # the headline
    ("whvi::fwht_rows_kernel<float, 11, 16, 0, false, true, 256, 1, false>", 128, "buffer_store_dwordx4", "spaced"),
    ("whvi::fwht_rows_kernel<float, 9, 16, 0, false, true, 256, 1, false>", 128, "buffer_store_dwordx4", "spaced"),
    ("whvi::fwht_rows_kernel<float, 11, 16, 0, false, true, 256, 1, true>", 128, "buffer_store_dwordx4", "spaced"),      # WHVI_FWHT_SIGNED_LANES
    ("whvi::fwht_rows_kernel<float, 9, 16, 0, false, true, 256, 1, true>", 128, "buffer_store_dwordx4", "spaced"),
    ("whvi::fwht_rows_kernel<__half, 12, 8, 2, false, true, 256, 1, false>", 128, "buffer_store_dwordx4", "spaced"),    # config 5
    ("whvi::fwht_rows_kernel<float, 13, 32, 0, false, true, 256, 1, false>", 168, "buffer_store_dwordx4", "spaced"),    # three waves per SIMD
    ("whvi::fused_shs_kernel<float, 11, 16, 1, false, true, 256, 0, 1, false, false>", 168, "buffer_store_dwordx4", "spaced"),   # config 3
    ("whvi::fused_shs_kernel<float, 12, 16, 1, false, true, 256, 0, 1, false, false>", 168, "buffer_store_dwordx4", "spaced"),
    ("whvi::wbar_fwd_kernel<float, 11, 16, true, false>", 128, "global_store_dwordx4", "adjacent"),
    ("whvi::wbar_fwd_kernel<float, 9, 16, true, false>", 128, "global_store_dwordx4", "adjacent"),
    ("whvi::diag_apply_kernel<float, 9, 4, false, true>", 64, None, None),                                              # config 2's layer (256 MiB: cached launch, quarter tiles, 8 waves per SIMD)
    ("whvi::diag_apply_kernel<float, 9, 16, true, true>", 128, "global_store_dwordx4", "adjacent"),                    # the same layer beyond the Infinity Cache
    ("whvi::diag_apply_kernel<float, 10, 16, true, false>", 128, "buffer_store_dwordx4", "spaced"),                     # config 4's middle layer
    ("whvi::stream_copy_kernel<float, 16, 256>", 128, "buffer_store_dwordx4", "spaced"),                                       # the measured ceiling
]


def _issue_violations(shipped, contracts):
    from shipped_isa import store_runs
    bad = []
    for name, budget, mnemonic, form in contracts:
        k = shipped.find(name)
        if k["scratch"] != 0:
            bad.append(f"{name}: {k['scratch']} B of scratch")
        if k["vgprs"] + k["agprs"] > budget:
            bad.append(f"{name}: {k['vgprs']} + {k['agprs']} registers > {budget} (occupancy budget)")
        if mnemonic is None:
            continue
        runs = store_runs(shipped.ops(name), mnemonic)
        if sum(runs) < 8:
            bad.append(f"{name}: no {mnemonic} stream found ({runs})")
        elif form == "spaced" and sum(r for r in runs if r > 1) > 6:       # (the compiler pairs the first two / last three)
            bad.append(f"{name}: {mnemonic} issued back to back in runs of {runs}")
        elif form == "adjacent" and max(runs) < 3:
            bad.append(f"{name}: {mnemonic} issued apart ({runs}); this write-only stream wants them back to back")
    return bad


def test_shipped_kernels_keep_their_occupancy_and_store_issue_form(tmp_path):
    """nothing enforced the instruction-issue rules the last 9 % rests on.  Reads the shipped library
    (no GPU, no recompilation): every kernel a PMC record names exists, none uses scratch, and the published
    instantiations keep their register budget and the issue pattern of their 16-byte stores.  The check is proven able to
    fail: the same kernel built with -DWHVI_ROWS_STORE_FORM=1 (chunk offsets as scalar offsets: the stores go out back to
    back, 5.89 instead of 6.44 TB/s) violates the headline's contract."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from shipped_isa import ShippedLibrary
    with ShippedLibrary() as shipped:
        assert len(shipped.kernels) > 1000
        assert [n for n, k in shipped.kernels.items() if k["scratch"]] == []
        records = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        for key, rec in records.items():
            if isinstance(rec, dict) and rec.get("kernel_symbol"):
                shipped.find(rec["kernel_symbol"])                      # KeyError (with near misses) when it is gone
        assert _issue_violations(shipped, ISSUE_CONTRACTS) == []
    # a tuning build of the headline's translation unit with the back-to-back store form: the contract must catch it
    obj = tmp_path / "fwht_f32_storeform1.o"
    build = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                            "-fvisibility=hidden", "-DWHVI_TUNING_BUILD", "-DWHVI_ROWS_STORE_FORM=1", "-c",
                            os.path.join(ROOT, "whvi_amd", "csrc", "fwht_f32.hip"), "-o", str(obj)],
                           capture_output=True, text=True, timeout=1200)
    assert build.returncode == 0, build.stderr[-2000:]
    with ShippedLibrary(str(obj)) as tuned:
        bad = _issue_violations(tuned, ISSUE_CONTRACTS[:1])
    assert len(bad) == 1 and "back to back" in bad[0], bad
