"""tools/_tuning.py -- measurement builds of the HIP library for the probes in this directory.

The shipped whvi_amd/libwhvi_hip.so reads no environment and carries no experimental instruction forms
(whvi_amd/csrc/tuning.hpp).  Probes that A/B a launch form through an environment switch (WHVI_FUSED_TUNE,
WHVI_WBAR_FWD_TILES, WHVI_STREAM_BIG_BLOCKS, ...) or a -D override call

    import _tuning; _tuning.use()                         # before importing whvi_amd
    _tuning.use("nopk", "-DWHVI_NO_PK")                   # a -D variant under its own tag

which builds whvi_amd/_exp/libwhvi_hip_<tag>.so with -DWHVI_TUNING_BUILD (make -C whvi_amd/csrc tuning) when it is
missing or older than the kernel sources and makes whvi_amd._hip load it (assigns its LIB_PATH: the shipped loader reads
no environment).  To run an unmodified script (bench.py, a probe) on such a build:

    python tools/_tuning.py --run whvi_amd/_exp/libwhvi_hip_<tag>.so bench.py --no-extras ...
  Build in the container: the GPU box has the
compiler too, but its minutes are better spent measuring."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "whvi_amd", "csrc")


def lib_path(tag: str = "tuning") -> str:
    return os.path.join(ROOT, "whvi_amd", "_exp", f"libwhvi_hip_{tag}.so")


def build(tag: str = "tuning", defs: str = "") -> str:
    out = lib_path(tag)
    sources = glob.glob(os.path.join(CSRC, "*.h*")) + [os.path.join(ROOT, "include", "whvi_hip.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(p) for p in sources):
        subprocess.check_call(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1)), "tuning", f"TAG={tag}",
                               f"DEFS={defs}"])
    return out


def load(path: str) -> str:
    """Make whvi_amd._hip load the library at ``path`` (must run before its first native call)."""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from whvi_amd import _hip
    if _hip._lib is not None and _hip.LIB_PATH != path:
        raise RuntimeError("_tuning.load: whvi_amd has already loaded " + _hip.LIB_PATH)
    _hip.LIB_PATH = os.path.abspath(path)
    return _hip.LIB_PATH


def use(tag: str = "tuning", defs: str = "") -> str:
    return load(build(tag, defs))


if __name__ == "__main__":
    import runpy
    import sys
    if len(sys.argv) >= 4 and sys.argv[1] == "--run":          # --run <lib.so> <script.py> [args...]
        load(sys.argv[2])
        sys.argv = sys.argv[3:]
        runpy.run_path(sys.argv[0], run_name="__main__")
    else:
        print(build(*(sys.argv[1:3])))
