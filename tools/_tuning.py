"""tools/_tuning.py -- measurement builds of the HIP library for the probes in this directory.

The shipped whvi_amd/libwhvi_hip.so reads no environment and carries no experimental instruction forms
(whvi_amd/csrc/tuning.hpp).  Probes that A/B a launch form through an environment switch (WHVI_FUSED_TUNE,
WHVI_WBAR_FWD_TILES, WHVI_STREAM_BIG_BLOCKS, ...) or a -D override call

    import _tuning; _tuning.use()                         # before importing whvi_amd
    _tuning.use("nopk", "-DWHVI_NO_PK")                   # a -D variant under its own tag

which builds whvi_amd/_exp/libwhvi_hip_<tag>.so with -DWHVI_TUNING_BUILD (make -C whvi_amd/csrc tuning) when it is
missing or older than the kernel sources and points WHVI_HIP_LIB at it.  Build in the container: the GPU box has the
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


def use(tag: str = "tuning", defs: str = "") -> str:
    path = build(tag, defs)
    os.environ["WHVI_HIP_LIB"] = path
    return path


if __name__ == "__main__":
    import sys
    print(build(*(sys.argv[1:3])))
