"""The reference's OWN unit tests (test/walsh.py, test/utils.py, test/networks.py, test/likelihoods.py -- SURVEY.md 2,
row 16: "consumers that must pass unchanged") run against this repo, in the two ways INTEGRATION.md 1 / 1b describes.
Only where the reference checkout exists (the build container); nothing of it is copied or written: the suites run in a
child process with the checkout as working directory and bytecode writing off.  Without a GPU the two `test_cuda_*`
cases cannot run (they are restated in tests/test_fwht_gpu.py::test_cuda_simple_and_large_like_reference); everything
else must pass."""
import os
import re
import subprocess
import sys

import pytest
import torch

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["test.walsh", "test.utils", "test.networks", "test.likelihoods"]

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "test")), reason="no reference checkout here")


# The suites run under a fixed torch seed: test/likelihoods.py::test_gaussian_vectorization_random compares an UNSEEDED
# random case against a sequential float32 loop with an absolute tolerance of 1e-4 on a value of ~420 -- its own loop's
# rounding noise fails it for 1 seed in 300 with the reference's implementation and 2 in 300 with this repo's (measured
# here); the launcher below only seeds the generator, the reference's test files run unmodified.
_SEEDED_UNITTEST = ("import sys, unittest, torch; torch.manual_seed(20240607); "
                    "unittest.main(module=None, argv=['unittest', '-v'] + sys.argv[1:])")


def _run(pythonpath, code=None):
    env = {**os.environ, "PYTHONPATH": pythonpath, "PYTHONDONTWRITEBYTECODE": "1"}
    cmd = [sys.executable, "-c", code] if code else [sys.executable, "-c", _SEEDED_UNITTEST, *SUITES]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REFERENCE, env=env)


def _check_suites(out):
    text = out.stderr + out.stdout
    ran = re.search(r"Ran (\d+) tests", text)
    assert ran and int(ran.group(1)) == 13, text[-3000:]
    bad = re.findall(r"^(?:ERROR|FAIL): (\w+)", text, flags=re.M)
    if torch.cuda.is_available():
        assert bad == [] and out.returncode == 0, text[-3000:]
    else:
        assert sorted(bad) == ["test_cuda_large", "test_cuda_simple"], text[-3000:]     # need a GPU, nothing else may fail
        assert text.count("No HIP GPUs are available") >= 2 or text.count("Torch not compiled with CUDA") >= 2, text[-3000:]


def test_reference_suites_with_only_the_native_modules_swapped():
    """`dropin/` on the path -- the reference's own `src` package (weights, layers, networks,
    likelihoods, utils, the three FWHT front-ends) on top of this repo's `fwht_cuda` / `fwht_cpp`."""
    path = os.path.join(ROOT, "dropin")
    where = _run(path, "import fwht_cuda, fwht_cpp, src.weights, src.fwht.cpp.fwht as f; "
                       "print(src.weights.__file__); print(fwht_cuda.__file__); print(fwht_cpp.__file__); print(f.__file__)")
    assert where.returncode == 0, where.stderr[-2000:]
    weights, cuda_mod, cpp_mod, front = where.stdout.split()
    assert weights.startswith(REFERENCE) and front.startswith(REFERENCE), "the reference's own src package must be the one in use"
    assert cuda_mod.startswith(path) and cpp_mod.startswith(path)
    _check_suites(_run(path))


def test_reference_suites_on_the_alias_package():
    """the repo root on the path -- `src.*` resolves to the alias package, i.e. to `whvi_amd`."""
    where = _run(ROOT, "import src.weights, src.layers; print(src.weights.__file__); print(src.layers.WHVILinear.__module__)")
    assert where.returncode == 0, where.stderr[-2000:]
    weights, module = where.stdout.split()
    assert weights.startswith(ROOT) and module.startswith("whvi_amd")
    _check_suites(_run(ROOT))


def test_reference_benchmark_script_runs_on_the_dropin_modules():
    """benchmarks/walsh.py is a consumer too -- dense H product against `FWHTFunction` of
    src/fwht/cpp on (1, D, D) inputs up to D = 8192 (benchmarks/walsh.py:16-42).  Must run to the end unchanged."""
    env = {**os.environ, "PYTHONPATH": os.path.join(ROOT, "dropin"), "PYTHONDONTWRITEBYTECODE": "1"}
    out = subprocess.run([sys.executable, "-m", "benchmarks.walsh"], capture_output=True, text=True, timeout=900,
                         cwd=REFERENCE, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("Evaluating benchmark_FWHT_speed_") == 4 and out.stdout.count("FWHTFunction_:") == 4, out.stdout


def test_reference_experiment_script_runs_on_the_alias_package(tmp_path):
    """The reference's UCI runner experiments/regression_experiments/run_yacht.py -- a consumer of
    ``src.evaluation.evaluate_bayesian_regression_dnn`` (src/evaluation.py:30-108) -- executed UNCHANGED against this repo:
    it loads ``../datasets/yacht_hydrodynamics.data`` (here a synthetic file of the data set's shape, 308 x 7), calls the
    protocol with the reference's positional signature and lets it write ``../checkpoints/yacht``.  Only the epoch counts
    are shortened (keyword defaults of this repo's function, bound before the script imports it); working directory and
    outputs are scratch, nothing is written into the checkout."""
    import numpy as np
    work = tmp_path / "experiments"
    work.mkdir()
    (tmp_path / "datasets").mkdir()
    rng = np.random.default_rng(0)
    np.savetxt(tmp_path / "datasets" / "yacht_hydrodynamics.data", rng.normal(size=(308, 7)))
    script = os.path.join(REFERENCE, "experiments", "regression_experiments", "run_yacht.py")
    code = ("import functools, runpy, src.evaluation as e; "
            "e.evaluate_bayesian_regression_dnn = functools.partial(e.evaluate_bayesian_regression_dnn, epochs1=1, epochs2=2, n_splits=2); "
            f"ns = runpy.run_path({script!r}, run_name='__main__'); "
            "print('RESULT', ns['error_mean'], ns['error_sd'], ns['mnll_mean'], ns['mnll_sd'], e.__file__)")
    env = {**os.environ, "PYTHONPATH": ROOT, "PYTHONDONTWRITEBYTECODE": "1"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=work, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][0].split()
    assert all(np.isfinite(float(v)) for v in line[1:5]) and line[5].startswith(ROOT)
    assert "Iteration 2/2" in out.stdout and "Using torch device" in out.stdout
    assert sorted(os.listdir(tmp_path / "checkpoints" / "yacht")) == ["iter-0", "iter-1"]
    assert os.listdir(tmp_path / "checkpoints" / "yacht" / "iter-1") == ["epoch-0.pth"]
