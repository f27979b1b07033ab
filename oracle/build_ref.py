"""oracle/build_ref.py -- TEST INFRASTRUCTURE.

Compiles the reference's own C++ FWHT (``<reference>/src/fwht/cpp/fwht.cpp``) from
where it lies into ``oracle/_ref/fwht_cpp<ext-suffix>.so`` with a direct ``g++``
command (not the reference's setup.py / jit.py).  The resulting module is importable
as ``fwht_cpp`` (``forward``/``backward``, src/fwht/cpp/fwht.cpp:31-34) and is used

  * by tests/ to prove the C restatement in fwht_oracle.c bit-equal to the reference,
  * by tests/golden/make_golden.py so the reference's Python package imports,
  * by bench.py's ``cpu_baseline`` leg (kind = "reference").

No reference source is copied: only the built binary lands in oracle/_ref/ (git-ignored).
Silently does nothing (exit 0) when the reference tree is absent -- the GPU box uses the
prebuilt file that travelled with the snapshot.
"""
import argparse
import os
import subprocess
import sys
import sysconfig


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args()

    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(args.reference, "src", "fwht", "cpp", "fwht.cpp")
    out_dir = os.path.join(here, "_ref")
    out = os.path.join(out_dir, "fwht_cpp" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(src):
        print(f"[build_ref] {src} not present; keeping any prebuilt {out}")
        return 0
    if os.path.exists(out) and not args.force and os.path.getmtime(out) >= os.path.getmtime(src):
        print(f"[build_ref] up to date: {out}")
        return 0

    import torch
    from torch.utils import cpp_extension as ce

    os.makedirs(out_dir, exist_ok=True)
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}"]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = [
        os.environ.get("CXX", "g++"), "-O2", "-DNDEBUG", "-fwrapv", "-std=c++17", "-fPIC", "-shared", "-w",  # distutils' own OPT flags
        "-DTORCH_EXTENSION_NAME=fwht_cpp", "-DTORCH_API_INCLUDE_EXTENSION_H",
        f"-D_GLIBCXX_USE_CXX11_ABI={abi}",
        *inc, src, "-o", out,
        f"-L{libdir}", f"-Wl,-rpath,{libdir}",
        "-ltorch_python", "-ltorch", "-ltorch_cpu", "-lc10",
    ]
    print("[build_ref]", " ".join(cmd))
    subprocess.check_call(cmd)
    print(f"[build_ref] built {out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
