"""Build recipe for a regular install -- the counterpart of the reference's two ``setup.py`` files
(src/fwht/cuda/setup.py:4-11 ``CUDAExtension('fwht_cuda', ...)``, src/fwht/cpp/setup.py ``CppExtension('fwht_cpp', ...)``):

    pip install --no-build-isolation .        # or: python setup.py build

runs ``make`` in ``whvi_amd/csrc`` (hipcc --offload-arch=gfx950 for ``libwhvi_hip.so``, g++ -fopenmp for
``libwhvi_cpu.so``) and installs the ``whvi_amd`` package with both libraries inside it plus the two top-level modules
the reference imports, ``fwht_cuda`` and ``fwht_cpp``.  The import-path alias package ``src/`` is NOT installed (a
top-level ``src`` in site-packages would shadow other projects); use it from the checkout.
Development, the tests and the benchmark use the in-tree build (``__graft_entry__.build()``) and need no install."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithNativeLibraries(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "whvi_amd", "csrc"), "-j", str(os.cpu_count() or 4)])
        super().run()


setup(
    name="whvi-amd",
    version="0.2.0",
    description="MI355X-native (gfx950) batched fast Walsh-Hadamard transform and WHVI weight-sample pipeline",
    packages=["whvi_amd", "whvi_amd.fwht"],
    py_modules=["fwht_cuda", "fwht_cpp"],
    package_data={"whvi_amd": ["libwhvi_hip.so", "libwhvi_cpu.so"]},
    python_requires=">=3.9",
    cmdclass={"build_py": BuildWithNativeLibraries},
)
