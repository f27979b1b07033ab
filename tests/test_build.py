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
