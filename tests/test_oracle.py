"""Pins the CPU oracle (oracle/) -- runs without a GPU.

The oracle is the checker for every GPU parity test, so it is itself checked against
  1. the reference's own known-answer vectors (test/walsh.py:12-13,17-18),
  2. the dense Hadamard identity the reference's tests use (test/walsh.py:22-49),
  3. golden vectors recorded from the live reference (tests/golden/*.npz, make_golden.py),
  4. the reference's own compiled C++ FWHT (oracle/_ref), bit for bit, when it has been built.
"""
import os

import numpy as np
import pytest

import oracle
from oracle import whvi_oracle as wo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def fg():
    return np.load(os.path.join(GOLD, "fwht_golden.npz"))


def test_reference_known_answer_vectors():
    # literal vectors of test/walsh.py:12-13 and :17-18
    a = np.array([[1.0, 2.0, 3.0, 4.0]], dtype=np.float32)
    assert np.array_equal(oracle.fwht(a), np.array([[10.0, -2.0, -4.0, 0.0]], dtype=np.float32))
    a = np.array([[0.0, 1.0, 2.0, 3.0]], dtype=np.float32)
    assert np.array_equal(oracle.fwht(a), np.array([[6.0, -2.0, -4.0, 0.0]], dtype=np.float32))


def test_known_answers_as_recorded_from_reference(fg):
    for k in ("kat1", "kat2"):
        assert np.array_equal(oracle.fwht(fg[k + "_in"]), fg[k + "_out"])


def test_dense_hadamard_identity():
    # test/walsh.py:22-49: D = 32, batch 1 and 40, atol 1e-5 against (H @ A.T).T
    rng = np.random.default_rng(0)
    H = oracle.hadamard(32)
    for batch in (1, 40):
        for _ in range(30):
            A = rng.standard_normal((batch, 32)).astype(np.float32)
            assert np.allclose(oracle.fwht(A), (H @ A.T.astype(np.float64)).T, atol=1e-5)
    assert np.array_equal(oracle.dense_wht(np.eye(8)), oracle.hadamard(8))


def test_hadamard_matches_reference_build_H(fg):
    assert np.array_equal(oracle.hadamard(8), fg["H_8"].astype(np.float64))


@pytest.mark.parametrize("d", [1, 2, 4, 8, 32, 64, 512, 1024, 4096])
def test_bit_equal_to_golden(fg, d):
    assert np.array_equal(oracle.fwht(fg[f"i32_in_{d}"]), fg[f"i32_out_{d}"])
    assert np.array_equal(oracle.fwht(fg[f"i32_in_{d}"].astype(np.float32)), fg[f"intf32_out_{d}"])
    assert np.array_equal(oracle.fwht(fg[f"i32_in_{d}"].astype(np.int64)), fg[f"i32_out_{d}"].astype(np.int64))
    f32 = oracle.fwht(fg[f"f32_in_{d}"])
    assert np.array_equal(f32.view(np.uint32), fg[f"f32_out_{d}"].view(np.uint32))
    assert np.array_equal(f32.view(np.uint32), fg[f"f32_pyout_{d}"].view(np.uint32))  # python FWHT is bit-equal too
    assert np.array_equal(oracle.fwht(fg[f"f64_in_{d}"]).view(np.uint64), fg[f"f64_out_{d}"].view(np.uint64))


def test_bit_equal_to_compiled_reference():
    ref = oracle.load_reference_cpp()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference; python oracle/build_ref.py)")
    import torch
    g = torch.Generator().manual_seed(3)
    for rows, d in ((19, 1024), (3, 4096), (7, 1), (5, 2), (40, 32)):
        x = torch.randn(rows, d, generator=g)
        assert np.array_equal(ref.forward(x).numpy().view(np.uint32), oracle.fwht(x.numpy()).view(np.uint32))
        xi = torch.randint(-8, 8, (rows, d), generator=g, dtype=torch.int32)
        assert np.array_equal(ref.forward(xi).numpy(), oracle.fwht(xi.numpy()))


def test_integer_overflow_wraps_like_the_reference():
    """Integer rows whose partial sums leave the type's range (inputs near +/-2^30 resp. +/-2^62, D = 4096: twelve
    doublings).  The reference's integer tensors wrap (ATen's two's complement add / sub under
    src/fwht/cpp/fwht.cpp:11-13); the oracle's integer legs are built with -fwrapv, agree with exact arithmetic reduced
    modulo 2^32 / 2^64, and -- when oracle/_ref is built -- with the reference's own compiled C++ FWHT on the same
    int32 / int64 tensors, bit for bit.  This is what tests/test_streaming_parity_gpu.py::test_int32_wraps_like_the_reference
    holds the HIP int32 kernels to."""
    rng = np.random.default_rng(2 ** 30)
    d, rows = 4096, 6
    mag = rng.integers((1 << 30) - 4096, (1 << 30) + 4096, (rows, d))
    x32 = (mag * rng.choice([-1, 1], (rows, d))).astype(np.int32)
    got32 = oracle.fwht(x32)
    exact = oracle.fwht(x32.astype(np.int64))                     # |.| <= 2^42: no overflow in 64 bits
    assert np.abs(exact).max() > 2 ** 31, "the inputs must make 32-bit sums overflow"
    assert got32.dtype == np.int32 and np.array_equal(got32, exact.astype(np.int32))
    # python integers (unbounded) on one row, reduced modulo 2^32
    row = [int(v) for v in x32[0]]
    h = 1
    while h < d:
        for i in range(0, d, 2 * h):
            for j in range(i, i + h):
                row[j], row[j + h] = row[j] + row[j + h], row[j] - row[j + h]
        h *= 2
    wrapped = np.array([((v + 2 ** 31) % 2 ** 32) - 2 ** 31 for v in row], dtype=np.int64)
    assert np.array_equal(got32[0].astype(np.int64), wrapped)
    x64 = ((mag.astype(np.int64) << 32) * rng.choice([-1, 1], (rows, d))).astype(np.int64)
    got64 = oracle.fwht(x64)
    assert np.array_equal(got64[:, 0], np.sum(x64.astype(np.uint64), axis=1, dtype=np.uint64).astype(np.int64))   # y[0] = sum, mod 2^64
    ref = oracle.load_reference_cpp()
    if ref is None:
        pytest.skip("oracle/_ref not built: wrap-around checked against exact arithmetic only")
    import torch
    assert np.array_equal(ref.forward(torch.from_numpy(x32)).numpy(), got32)
    assert np.array_equal(ref.forward(torch.from_numpy(x64)).numpy(), got64)


def test_stage_order_tolerance_statement():
    """SURVEY.md finding 3: descending strides (the reference CUDA kernel's order) differ from the
    ascending order only by fp32 rounding, a few 1e-7 * max|y|; identical on integers."""
    rng = np.random.default_rng(1)
    for d in (512, 4096):
        x = rng.standard_normal((8, d)).astype(np.float32)
        up, down = oracle.fwht(x), oracle.fwht_descending(x)
        assert np.abs(up - down).max() <= 1e-6 * np.abs(up).max()
        xi = rng.integers(-8, 8, (8, d)).astype(np.float32)
        assert np.array_equal(oracle.fwht(xi), oracle.fwht_descending(xi))


def test_pipeline_is_composition_of_primitives():
    """oracle.pipeline == matmul_diag_{left,right} o fwht o ... (src/utils.py:4-23, src/weights.py:73)."""
    rng = np.random.default_rng(2)
    S, G, D = 3, 8, 8
    x = rng.standard_normal((S * G, D)).astype(np.float32)
    a, c = rng.standard_normal(G).astype(np.float32), rng.standard_normal(G).astype(np.float32)
    b = rng.standard_normal((S, G)).astype(np.float32)
    got = oracle.pipeline(x, a, b, c, n_samples=S, sample_stride=G, group_rows=G, axis="row")
    for s in range(S):
        blk = x[s * G:(s + 1) * G]
        want = a[:, None] * oracle.fwht((b[s][:, None] * oracle.fwht((c[:, None] * blk).astype(np.float32))).astype(np.float32))
        assert np.array_equal(got[s * G:(s + 1) * G], want.astype(np.float32))
    # column axis, (batch, sample, D) row order
    B = 4
    x = rng.standard_normal((B * S, D)).astype(np.float32)
    a, c = rng.standard_normal(D).astype(np.float32), rng.standard_normal(D).astype(np.float32)
    b = rng.standard_normal((S, D)).astype(np.float32)
    got = oracle.pipeline(x, a, b, c, n_samples=S, sample_stride=1, axis="col")
    for r in range(B * S):
        want = a * oracle.fwht((b[r % S] * oracle.fwht((c * x[r])[None].astype(np.float32))[0])[None].astype(np.float32))[0]
        assert np.array_equal(got[r], want.astype(np.float32))
    # the dense matrix it stands for: y = x S2 H G H S1 (column scaling), in float64
    H = oracle.hadamard(D)
    x64 = x.astype(np.float64)
    got64 = oracle.pipeline(x64, a, b, c, n_samples=S, sample_stride=1, axis="col")
    for r in range(B * S):
        M = np.diag(c.astype(np.float64)) @ H @ np.diag(b[r % S].astype(np.float64)) @ H @ np.diag(a.astype(np.float64))
        assert np.allclose(got64[r], x64[r] @ M, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("axis", ["col", "row"])
def test_per_sample_pipeline_is_the_shared_pipeline_sample_by_sample(axis, dtype):
    """oracle.pipeline(a_per_sample, c_per_sample) -- the checker of whvi_fused_shs_ex's WHVI_FUSED_{A,C}_PER_SAMPLE flags,
    i.e. batches of independent weight matrices (src/weights.py:130-132,179-180) -- equals the shared-vector pipeline
    (pinned above) applied row by row with that row's sample's vectors; both row orders."""
    rng = np.random.default_rng(5)
    d, S, B, G = 16, 3, 4, 5
    rows = B * S * G if axis == "row" else B * S
    unit = G if axis == "row" else d
    x = rng.standard_normal((rows, d)).astype(dtype)
    a, b, c = (rng.standard_normal((S, unit)).astype(dtype) for _ in range(3))
    for stride in (1, G if axis == "row" else B):
        y = oracle.pipeline(x, a, b, c, n_samples=S, sample_stride=stride, group_rows=G, axis=axis, a_per_sample=True,
                            c_per_sample=True)
        for r in range(rows):
            s = (r // stride) % S
            if axis == "col":
                w = oracle.pipeline(x[r:r + 1], a[s], b[s][None], c[s], n_samples=1, axis="col")[0]
            else:
                rr = r % G
                w = oracle.pipeline(x[r:r + 1], a[s][rr:rr + 1], b[s][rr:rr + 1], c[s][rr:rr + 1], n_samples=1,
                                    group_rows=1, axis="row")[0]
            assert np.array_equal(y[r], w), (axis, stride, r)
        only_a = oracle.pipeline(x, a, b, c[0], n_samples=S, sample_stride=stride, group_rows=G, axis=axis, a_per_sample=True)
        shared = oracle.pipeline(x, a[0], b, c[0], n_samples=S, sample_stride=stride, group_rows=G, axis=axis)
        first = ((np.arange(rows) // stride) % S) == 0
        assert np.array_equal(only_a[first], shared[first])


def test_w_bar_collapses_to_diagonal_exactly():
    """SURVEY.md finding 1: with butterflies, w_bar(u) == diag(s1 * (D * (u * s2))) bit for bit."""
    rng = np.random.default_rng(4)
    for D in (8, 512):
        s1, s2, u = (rng.standard_normal(D).astype(np.float32) for _ in range(3))
        W = wo.w_bar(s1, s2, u)
        assert np.array_equal(W, np.diag(s1 * (np.float32(D) * (u * s2))).astype(np.float32))


def _bundles():
    g = np.load(os.path.join(GOLD, "whvi_golden.npz"))
    names = sorted({k.split("/", 1)[0] for k in g.files})
    return g, names


@pytest.mark.parametrize("name", ["sq8", "sq64b", "sq512", "sq4096", "st3x16", "st5x7b", "st13x128",
                                  "col1x10b", "col16x1"])
def test_whvi_oracle_matches_reference_bundles(name):
    """Forward output and KL of the numpy restatement vs bundles recorded from the live reference.
    Tolerance 1e-5 relative to max|y| (BASELINE.json north_star): the reference's host path for
    D < 4096 goes through a dense H matmul (src/weights.py:38-39) whose rounding noise the exact
    butterfly does not have (measured <= 3e-6 at D = 512)."""
    g, _ = _bundles()
    b = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(name + "/")}
    params = {k[len("param."):]: v for k, v in b.items() if k.startswith("param.")}
    layer = wo.layer_from_params(int(b["n_in"]), int(b["n_out"]), float(b["lambda_"]), params)
    eps = [b[f"eps{i}"] for i in range(int(b["n_eps"]))]
    y = layer.forward(b["x"], eps if isinstance(layer, wo.Stacked) else eps[0])
    assert y.shape == b["y"].shape
    assert np.abs(y - b["y"]).max() <= 1e-5 * np.abs(b["y"]).max()
    assert abs(float(layer.kl) - float(b["kl"])) <= 1e-5 * abs(float(b["kl"]))


def test_sample_and_w_bar_vs_reference():
    g, _ = _bundles()
    b = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith("sample16/")}
    sq = wo.Square(b["s1"], b["s2"], b["g_mu"], b["g_rho"], 0.7)
    for got, want in ((sq.sample(b["eps0"]), b["W"]), (wo.w_bar(b["s1"], b["s2"], b["u"]), b["w_bar"])):
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
        # the reference's off-diagonals are dense-matmul rounding noise, ours are exact zeros
        assert np.count_nonzero(got - np.diag(np.diag(got))) == 0


@pytest.mark.parametrize("args,expect", [((3, 16), (4, 16, 1, 4)), ((5, 7), (8, 8, 3, 1)),
                                         ((13, 128), (16, 128, 3, 8)), ((3, 1024), (4, 1024, 1, 256)),
                                         ((8, 8), (8, 8, 0, 1)), ((4, 6), (4, 8, 0, 2))])
def test_setup_dimensions_probes(args, expect):
    assert wo.setup_dimensions(*args) == expect   # values probed on the reference


def _pipeline_golden():
    return np.load(os.path.join(GOLD, "pipeline_golden.npz"))


@pytest.mark.parametrize("key", ["f32_D64", "f32_D512", "f32_D2048", "f64_D64", "f64_D512"])
def test_column_pipeline_vs_vectors_recorded_from_the_reference(key):
    """The column-scaling pipeline (BASELINE config 3's dataflow) against vectors composed by the LIVE reference from its
    own ``matmul_diag_right`` (src/utils.py:15-23) and FWHT function (src/fwht/cpp/fwht.py:7-18) --
    tests/golden/make_golden_r3.py: shared and per-sample outer vectors, both row orders, and the one-transform half,
    bit for bit.  This is what pins ``oracle.pipeline(axis="col")`` -- the checker of every fused-kernel GPU test --
    to the reference's code rather than to the oracle's own composition."""
    g = _pipeline_golden()
    x, s1, s2, gk = (g[f"{key}/{n}"] for n in ("x", "s1", "s2", "g"))
    S, B = gk.shape[0], x.shape[0] // gk.shape[0]
    for order, stride in (("batch", 1), ("sample", B)):
        kw = dict(n_samples=S, sample_stride=stride, axis="col")
        got = oracle.pipeline(x, s1[0], gk, s2[0], **kw)
        assert np.array_equal(got.view(np.uint8), g[f"{key}/{order}/shared"].view(np.uint8)), (key, order)
        got = oracle.pipeline(x, s1, gk, s2, a_per_sample=True, c_per_sample=True, **kw)
        assert np.array_equal(got.view(np.uint8), g[f"{key}/{order}/per_sample"].view(np.uint8)), (key, order)
        # the one-transform half a * fwht(b * x): the oracle's plain transform between two multiplies
        smp = (np.arange(x.shape[0]) // stride) % S
        half = (s1[0] * oracle.fwht((gk[smp] * x).astype(x.dtype))).astype(x.dtype)
        assert np.array_equal(half.view(np.uint8), g[f"{key}/{order}/one_transform"].view(np.uint8)), (key, order)


def test_integer_wrap_vectors_recorded_from_the_reference():
    g = _pipeline_golden()
    for D in (64, 4096):
        for kind in ("i32", "i64"):
            assert np.array_equal(oracle.fwht(g[f"wrap_{kind}_D{D}/in"]), g[f"wrap_{kind}_D{D}/out"]), (kind, D)
