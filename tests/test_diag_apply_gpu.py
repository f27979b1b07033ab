"""whvi_diag_apply / whvi_diag_apply_bwd (whvi_amd/csrc/diag_apply.hpp): `h @ (w_bar(g_mu) + w_bar(g_sigma eps_k)).T + bias`
of src/weights.py:87-93,101-102 for all MC samples in one launch, without the matrices.

Checked against (a) the numpy oracle's Square layer (oracle/whvi_oracle.py: the reference's dataflow -- two w_bar matrices
through the C butterfly oracle, their sum, a dense product), (b) this library's as-written route (weight construction kernel +
rocBLAS GEMM) on every shape class of the dispatch, VALUE-identical (`==`, i.e. zeros compare by value: include/whvi_hip.h),
(c) the same route on NON-FINITE inputs -- inf / NaN in the activations, non-finite and overflowing parameters -- where the
matrix route turns other outputs into NaN, and (d) its backward against the matrix route's autograd and float64 gradcheck."""
import numpy as np
import pytest
import torch

from oracle import whvi_oracle as wo
from whvi_amd import _hip
from whvi_amd.weights import DiagApplyFunction, WBarFunction, WHVISquarePow2Matrix

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _operands(dtype, D, S, seed, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    kw = dict(device=DEV, dtype=dtype, generator=g)
    s1, s2 = torch.randn(D, **kw) * scale, torch.randn(D, **kw) * scale
    u = torch.randn(1 + S, D, **kw)
    bias = torch.randn(1, D, **kw)
    return s1, s2, u, bias, g


def matrix_route(x, s1, s2, u, bias, mean_plus=True):
    """The as-written route on the GPU: W_k = w_bar(u_0) + w_bar(u_{1+k}) (weight-construction kernel, real butterflies),
    dense product, bias."""
    W = WBarFunction.apply(s1.unsqueeze(0), u.unsqueeze(0), s2.unsqueeze(0), None, mean_plus).squeeze(0)    # (S, D, D)
    out = torch.matmul(x, W.transpose(1, 2))
    return out + bias if bias is not None else out


def same_values(a, b):
    """== everywhere (zeros by value), NaN exactly where the other has NaN."""
    na, nb = torch.isnan(a), torch.isnan(b)
    return bool((na == nb).all()) and bool((a[~na] == b[~nb]).all())


@pytest.mark.parametrize("D,S,B", [(8, 3, 5), (64, 2, 7), (512, 3, 9)])
def test_vs_numpy_oracle_layer(D, S, B, hip_lib):
    """Sample by sample against oracle.Square.forward: the reference's sample_lrt + bias (src/weights.py:87-102)."""
    s1, s2, _, bias, g = _operands(torch.float32, D, S, 11 * D + S, 0.3)
    g_mu = torch.randn(D, device=DEV, generator=g) * 0.2
    g_rho = torch.rand(D, device=DEV, generator=g) - 3
    eps = torch.randn(S, D, device=DEV, generator=g)
    x = torch.randn(B, D, device=DEV, generator=g)
    u, _, _ = _hip.reparam_kl(g_mu.unsqueeze(0), g_rho.unsqueeze(0), eps.unsqueeze(0), 1e-5)
    got = _hip.diag_apply(x, s1, s2, u[0], bias, n_samples=S)
    sq = wo.Square(s1.cpu().numpy(), s2.cpu().numpy(), g_mu.cpu().numpy(), g_rho.cpu().numpy(), 1e-5, bias.cpu().numpy())
    # the oracle's softplus is numpy's log1p(exp(.)): feed it the GPU's sigma through eps' = sigma_gpu * eps / sigma_np is not
    # exact -- instead pin the reparameterisation separately (tests/test_fused_gpu.py) and give the oracle u directly
    un = u[0].cpu().numpy()
    xn = x.cpu().numpy()
    for k in range(S):
        W = (wo.w_bar(sq.s1, sq.s2, un[0]) + wo.w_bar(sq.s1, sq.s2, un[1 + k])).astype(np.float32)
        want = (xn @ W.T).astype(np.float32) + sq.bias
        assert np.array_equal(got[k].cpu().numpy(), want), k
        off = W.copy()
        np.fill_diagonal(off, 0)
        assert np.abs(off).max() == 0.0                                         # SURVEY finding 1: exactly diagonal


SHAPES = [(torch.float32, 4, 5, 37), (torch.float32, 16, 3, 129), (torch.float32, 128, 4, 33), (torch.float32, 256, 3, 70),
          (torch.float32, 512, 32, 64), (torch.float32, 512, 5, 61), (torch.float32, 1024, 3, 47), (torch.float32, 2048, 2, 19),
          (torch.float32, 4096, 3, 9), (torch.float64, 2, 3, 50), (torch.float64, 64, 3, 21), (torch.float64, 512, 2, 40),
          (torch.float64, 2048, 2, 11)]


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("dtype,D,S,B", SHAPES)
def test_value_identical_to_the_matrix_route(dtype, D, S, B, shared, hip_lib):
    """Every tile geometry (rows per tile 1 .. 1024, rows shorter / longer than a wave's 64 chunks), batches that do and
    do not fill whole blocks (blocks that straddle two samples take the per-chunk path), a partial last tile, shared
    and per-sample inputs, with and without bias and mean row."""
    s1, s2, u, bias, g = _operands(dtype, D, S, 7 * D + S)
    x = torch.randn((B, D) if shared else (S, B, D), device=DEV, dtype=dtype, generator=g)
    keep = x.clone()
    x.view(-1, D)[1, : D // 2] = 0.0                              # exact zeros of both signs among the activations (a ReLU's output):
    x.view(-1, D)[2, : D // 2] = -0.0                             # products of -0 whose sign the GEMM's accumulation drops
    keep = x.clone()
    got = _hip.diag_apply(x, s1, s2, u, bias, n_samples=S)
    assert _hip.last_kernel().startswith("whvi::diag_apply_kernel<")
    want = matrix_route(x, s1, s2, u, bias)
    assert got.shape == (S, B, D) and torch.equal(got, want) and torch.equal(x, keep)
    ibits = torch.int32 if dtype == torch.float32 else torch.int64
    nobias, nobias_want = _hip.diag_apply(x, s1, s2, u, None, n_samples=S), matrix_route(x, s1, s2, u, None)
    assert bool((nobias == 0).any()) and torch.equal(nobias.view(ibits), nobias_want.view(ibits)), "bits, zeros included"
    assert torch.equal(_hip.diag_apply(x, s1, s2, u, None, n_samples=S), matrix_route(x, s1, s2, u, None))
    # direct weight sampling (src/weights.py:104-108): one w_bar per sample, no mean row
    assert torch.equal(_hip.diag_apply(x, s1, s2, u[1:], bias, n_samples=S, mean_plus=False),
                       matrix_route(x, s1, s2, u[1:], bias, mean_plus=False))
    # the launch forms a size-based dispatch would not pick here: streaming, and the 16 KiB tiles of streams at a cached size
    for tune in (_hip.DIAG_TUNE_NT, _hip.DIAG_TUNE_CACHED | 128, _hip.DIAG_TUNE_NT | _hip.DIAG_TUNE_PLAIN_ORDER):
        assert torch.equal(_hip.diag_apply(x, s1, s2, u, bias, n_samples=S, tune=tune), want), tune
    if not shared:                                                               # in place
        assert torch.equal(_hip.diag_apply(x, s1, s2, u, bias, n_samples=S, out=x), want)


@pytest.mark.parametrize("dtype,D,S,B", [(torch.float32, 512, 32, 4096 + 3), (torch.float32, 1024, 5, 16384 + 1)])
def test_streaming_size_vs_the_matrix_route_on_sampled_rows(dtype, D, S, B, hip_lib):
    """> 256 MiB written (the non-temporal launch, XCD-contiguous block order): BASELINE config 2's shape with a ragged
    batch, shared and per-sample input; sampled rows against the dense product with the sample's matrix."""
    s1, s2, u, bias, g = _operands(dtype, D, S, D + S)
    for shared in (True, False):
        x = torch.randn((B, D) if shared else (S, B, D), device=DEV, dtype=dtype, generator=g)
        got = _hip.diag_apply(x, s1, s2, u, bias, n_samples=S)
        assert got.numel() * got.element_size() > (256 << 20)
        assert _hip.last_kernel().endswith(f", true, {'true' if shared else 'false'}>"), _hip.last_kernel()
        rng = np.random.default_rng(B)
        bs = torch.from_numpy(np.unique(np.concatenate([[0, 1, 31, 32, B - 2, B - 1], rng.integers(0, B, 40)]))).to(DEV)
        W = WBarFunction.apply(s1.unsqueeze(0), u.unsqueeze(0), s2.unsqueeze(0), None, True).squeeze(0)
        for k in (0, 1, S // 2, S - 1):
            xs = x[bs] if shared else x[k][bs]
            assert torch.equal(got[k][bs], xs @ W[k].T + bias), (shared, k)
        assert bool(torch.isfinite(got[:, ::257]).all())
        del got, x


def _poison(x, g, n_rows, kinds=(float("inf"), float("-inf"), float("nan"))):
    """Put non-finite values into ``n_rows`` random rows of x (last dim = the row): one per row for the first half of them,
    two or three for the rest; returns the flat row indices."""
    flat = x.view(-1, x.shape[-1])
    rows = torch.randperm(flat.shape[0], device=DEV, generator=g)[:n_rows]
    for n, r in enumerate(rows.tolist()):
        for _ in range(1 if n < n_rows // 2 else 2 + n % 2):
            c = int(torch.randint(0, x.shape[-1], (1,), device=DEV, generator=g))
            flat[r, c] = kinds[(n + c) % len(kinds)]
    return rows


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("dtype,D,S,B", [(torch.float32, 4, 3, 200), (torch.float32, 64, 3, 50), (torch.float32, 512, 4, 40),
                                         (torch.float32, 1024, 2, 33), (torch.float32, 4096, 2, 12), (torch.float64, 256, 3, 30),
                                         (torch.float64, 2048, 2, 9)])
def test_non_finite_activations_propagate_like_the_dense_product(dtype, D, S, B, shared, hip_lib):
    """inf / -inf / NaN in h: the dot products of the matrix route meet W's exact zeros (inf * 0 = NaN), so every OTHER output
    of such a row is NaN and the output at the column itself is the plain product (src/weights.py:93 on a non-finite h)."""
    s1, s2, u, bias, g = _operands(dtype, D, S, 3 * D + S)
    x = torch.randn((B, D) if shared else (S, B, D), device=DEV, dtype=dtype, generator=g)
    x.view(-1, D)[0, 0] = 0.0                                       # a zero next to ...
    rows = _poison(x, g, max(4, x.numel() // D // 5))
    x.view(-1, D)[rows[0], D - 1] = float("inf")                   # ... and the edge columns of a row
    x.view(-1, D)[rows[1], 0] = float("-inf")
    clean = [r for r in range(x.numel() // D) if r not in set(rows.tolist())][-1]
    x.view(-1, D)[clean, D // 2] = float("inf")                    # exactly one: its own output stays +/-inf
    got = _hip.diag_apply(x, s1, s2, u, bias, n_samples=S)
    want = matrix_route(x, s1, s2, u, bias)
    assert same_values(got, want)
    assert bool(torch.isnan(got).any()) and bool(torch.isinf(got).any()) and bool(torch.isfinite(got).any())


@pytest.mark.parametrize("dtype,D", [(torch.float32, 8), (torch.float32, 512), (torch.float32, 2048), (torch.float64, 128)])
def test_non_finite_and_overflowing_parameters_like_the_matrix_route(dtype, D, hip_lib):
    """s1_i = inf / NaN (s1_i * 0 among the off-diagonals), u_i s2_i = inf / NaN, and finite values whose partial sums
    2^k u_i s2_i overflow inside the second transform (inf - inf) poison ROW i of W, hence output column i; a product that
    only overflows at the last stage (D u_i s2_i = inf, D/2 u_i s2_i finite) gives +/-inf on the diagonal and clean zeros."""
    S, B = 3, 17
    s1, s2, u, bias, g = _operands(dtype, D, S, 5 * D)
    big = torch.finfo(dtype).max
    s1[1] = float("inf")
    s1[2] = float("nan")
    u[1, 3] = float("inf")
    u[0, 4] = float("nan")
    s2[5] = float("-inf")
    u[2, 6], s2[6] = big / D * 1.5, 1.0              # D/2 * v finite, D * v = inf: a clean inf on the diagonal (sample 1)
    u[2, 7], s2[7] = big / D * 3.0, 1.0              # D/2 * v overflows: NaNs in row 7 of W (sample 1)
    u[0, 0], s2[0] = big / 4, -1.0                   # the mean row: every sample
    x = torch.randn(S, B, D, device=DEV, dtype=dtype, generator=g)
    x[0, 0, 6] = 0.0                                 # 0 * inf
    got = _hip.diag_apply(x, s1, s2, u, bias, n_samples=S)
    want = matrix_route(x, s1, s2, u, bias)
    assert same_values(got, want)
    assert bool(torch.isinf(got[1, 1:, 6]).all()) and bool(torch.isnan(got[1, :, 7]).all()) and bool(torch.isfinite(got[0, :, 7]).all())
    if D >= 16:
        assert bool(torch.isfinite(got[..., 8:]).all())


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("dtype,D,S,B", [(torch.float32, 4, 3, 300), (torch.float32, 128, 3, 77), (torch.float32, 512, 4, 130),
                                         (torch.float32, 1024, 3, 65), (torch.float32, 4096, 2, 21), (torch.float64, 512, 2, 40)])
def test_backward_vs_the_matrix_route_autograd(dtype, D, S, B, shared, hip_lib):
    """One-call backward against autograd through weight construction + GEMM (whvi_wbar_bwd, rocBLAS): north_star's 1e-5
    relative for float32 (the matrix route's own backward carries the rounding noise of D^2-term butterfly sums that
    cancel down to one term, whvi_wbar_bwd; the closed form here has none), 1e-12 for float64."""
    s1, s2, u, bias, g = _operands(dtype, D, S, 13 * D + S, 0.5)
    x = torch.randn((B, D) if shared else (S, B, D), device=DEV, dtype=dtype, generator=g)
    gout = torch.randn(S, B, D, device=DEV, dtype=dtype, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, s1, s2, u, bias)]
    got = torch.autograd.grad(DiagApplyFunction.apply(*leaves, S, True), leaves, gout)
    # (the launch note is per thread and autograd ran the backward on its own: name the kernel through a direct call)
    direct = _hip.diag_apply_bwd(gout, x, s1, s2, u, n_samples=S)
    assert _hip.last_kernel().startswith("whvi::diag_apply_bwd_kernel<") and _hip.last_kernel().endswith(", true>")
    assert torch.equal(direct[0] if not shared else direct[0].sum(dim=0), got[0])
    ref_leaves = [t.clone().requires_grad_(True) for t in (x, s1, s2, u, bias)]
    want = torch.autograd.grad(matrix_route(*ref_leaves), ref_leaves, gout)
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    for name, a, b in zip(("x", "s1", "s2", "u", "bias"), got, want):
        assert a.shape == b.shape, name
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), name
    # float64 closed form of the same expression: the kernel's float32 sums stay within float32 summation noise of it
    if dtype == torch.float32:
        d64 = [t.double().requires_grad_(True) for t in (x, s1, s2, u, bias)]
        exact = torch.autograd.grad(DiagApplyFunction._reference_ops(*d64, True), d64, gout.double())
        for name, a, b in zip(("x", "s1", "s2", "u", "bias"), got, exact):
            assert float((a.double() - b).abs().max()) <= 2e-6 * float(b.abs().max()), name
    # no gradient wanted for the input: the kernel skips writing it
    l2 = [t.clone().requires_grad_(i > 0) for i, t in enumerate((x, s1, s2, u, bias))]
    again = torch.autograd.grad(DiagApplyFunction.apply(*l2, S, True), l2[1:], gout)
    assert _hip.diag_apply_bwd(gout, x, s1, s2, u, n_samples=S, need_grad_x=False)[0] is None and _hip.last_kernel().endswith(", false>")
    for a, b in zip(again, got[1:]):
        assert torch.equal(a, b)


def test_backward_is_deterministic_and_gradcheck_float64(hip_lib):
    D, S, B = 16, 2, 5
    s1, s2, u, bias, g = _operands(torch.float64, D, S, 99, 0.7)
    for shared in (False, True):
        x = torch.randn((B, D) if shared else (S, B, D), device=DEV, dtype=torch.float64, generator=g)
        leaves = [t.clone().requires_grad_(True) for t in (x, s1, s2, u, bias)]
        fn = lambda *a: DiagApplyFunction.apply(*a, S, True)       # noqa: E731
        assert torch.autograd.gradcheck(fn, leaves, eps=1e-6, atol=1e-7)
        assert torch.autograd.gradgradcheck(fn, leaves, eps=1e-6, atol=1e-6)
        fn1 = lambda x_, a, c, u_: DiagApplyFunction.apply(x_, a, c, u_, None, S, False)       # noqa: E731
        assert torch.autograd.gradcheck(fn1, [leaves[0], leaves[1], leaves[2], leaves[3][1:].detach().requires_grad_(True)],
                                        eps=1e-6, atol=1e-7)
    # summation order is fixed: two runs, the same bits
    s1, s2, u, bias, g = _operands(torch.float32, 512, 8, 5)
    x = torch.randn(8, 1000, 512, device=DEV, generator=g)
    gout = torch.randn(8, 1000, 512, device=DEV, generator=g)
    a = _hip.diag_apply_bwd(gout, x, s1, s2, u, n_samples=8)
    b = _hip.diag_apply_bwd(gout, x, s1, s2, u, n_samples=8)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1][:, 1:], b[1][:, 1:])


def test_module_routes(monkeypatch, hip_lib):
    """WHVISquarePow2Matrix: "auto" = the one-launch diagonal on the GPU, ``faithful_dataflow`` = weight construction +
    GEMM; same eps stream, same values; forward / forward_mc / direct sampling, with bias; shapes the kernel lacks (D = 2)
    fall back to the matrix route."""
    torch.manual_seed(3)
    sq = WHVISquarePow2Matrix(256, bias=True).to(DEV)
    with torch.no_grad():
        sq.g_mu.normal_()
        sq.bias.normal_()
    x = torch.randn(40, 256, device=DEV)
    assert sq.exploit_diagonal is None and not sq.faithful_dataflow

    def run(fn):
        torch.manual_seed(5)
        return fn()

    for fn in (lambda: sq(x), lambda: sq(x, use_lrt=False), lambda: sq.forward_mc(x, 6),
               lambda: sq.forward_mc(x.expand(6, -1, -1).contiguous(), 6), lambda: sq(x.view(4, 10, 256))):
        sq.faithful_dataflow = False
        fast = run(fn)
        assert "diag_apply_kernel" in _hip.last_kernel()
        sq.faithful_dataflow = True
        assert sq.exploit_diagonal is False
        slow = run(fn)
        assert torch.equal(fast, slow)
    sq.faithful_dataflow = False
    assert sq.exploit_diagonal == "auto"
    monkeypatch.setattr(WHVISquarePow2Matrix, "default_exploit_diagonal", False)
    assert WHVISquarePow2Matrix(8).faithful_dataflow
    monkeypatch.undo()
    tiny = WHVISquarePow2Matrix(2).to(DEV)
    out = tiny.forward_mc(torch.randn(5, 2, device=DEV), 3)
    assert out.shape == (3, 5, 2) and "diag_apply" not in _hip.last_kernel()


def test_argument_checks(hip_lib):
    s1, s2, u, bias, g = _operands(torch.float32, 64, 2, 1)
    x = torch.randn(2, 5, 64, device=DEV)
    with pytest.raises(RuntimeError, match="outside the supported range"):
        _hip.diag_apply(torch.randn(2, 5, 8192, device=DEV), torch.randn(8192, device=DEV), torch.randn(8192, device=DEV),
                        torch.randn(3, 8192, device=DEV), n_samples=2)
    with pytest.raises(RuntimeError, match="operand shapes"):
        _hip.diag_apply(x, s1, s2, u[1:], n_samples=2)
    with pytest.raises(RuntimeError, match="must be"):
        _hip.diag_apply(x[0, 0], s1, s2, u, n_samples=2)
    fn = _hip.lib().whvi_diag_apply_f32
    p = x.data_ptr()
    assert fn(p, p, s1.data_ptr(), s2.data_ptr(), u.data_ptr(), None, 2, 5, 6, 1, None) == -5        # shared input, in place
    assert fn(p, p + 16, s1.data_ptr(), s2.data_ptr(), u.data_ptr(), None, 2, 4, 6, 0, None) == -5   # partial overlap
    assert fn(p, p, s1.data_ptr(), s2.data_ptr(), u.data_ptr(), None, 2, 5, 6, 1024, None) == -1     # unknown flag
    assert fn(p, p, s1.data_ptr(), s2.data_ptr(), u.data_ptr(), None, 2, 5, 13, 0, None) == -2
    assert fn(None, None, None, None, None, None, 0, 5, 6, 0, None) == 0                             # nothing to do
    bw = _hip.lib().whvi_diag_apply_bwd_f32
    assert bw(None, p, p, p, p, s1.data_ptr(), s2.data_ptr(), u.data_ptr(), None, 2, 5, 6, 7, 0, None) == -1      # n_slabs > B
    assert "slab" in _hip.last_error()


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("relu_in,relu_out", [(True, False), (False, True), (True, True)])
@pytest.mark.parametrize("dtype,D,S,B", [(torch.float32, 16, 3, 70), (torch.float32, 512, 4, 130), (torch.float32, 1024, 3, 65),
                                         (torch.float64, 256, 2, 33)])
def test_fused_relu_neighbours_equal_separate_passes(dtype, D, S, B, relu_in, relu_out, shared, hip_lib):
    """WHVI_DIAG_RELU_IN / _OUT: ``relu(layer(relu(x)))`` in the one launch == torch.relu passes around the unfused launch
    (== around the matrix route), forward values -- non-finite activations included: relu(-inf) = 0 is finite, relu(NaN)
    = NaN poisons its row -- and every gradient, bit for bit (the masks are recomputed with the forward's roundings)."""
    s1, s2, u, bias, g = _operands(dtype, D, S, 17 * D + S, 0.6)
    x = torch.randn((B, D) if shared else (S, B, D), device=DEV, dtype=dtype, generator=g)
    flat = x.view(-1, D)
    flat[3, 1], flat[5, 2], flat[7, 0], flat[9, D - 1] = float("-inf"), float("nan"), float("inf"), 0.0
    pre = torch.relu(x) if relu_in else x
    want = matrix_route(pre, s1, s2, u, bias)
    want = torch.relu(want) if relu_out else want
    got = _hip.diag_apply(x, s1, s2, u, bias, n_samples=S, relu_in=relu_in, relu_out=relu_out)
    assert same_values(got, want)
    unfused = _hip.diag_apply(pre, s1, s2, u, bias, n_samples=S)
    assert same_values(got, torch.relu(unfused) if relu_out else unfused)
    # gradients (finite data): the fused Function against torch.relu around the unfused Function
    x = torch.randn((B, D) if shared else (S, B, D), device=DEV, dtype=dtype, generator=g)
    gout = torch.randn(S, B, D, device=DEV, dtype=dtype, generator=g)
    for with_bias in (True, False):
        b = bias if with_bias else None
        la = [t.clone().requires_grad_(True) for t in (x, s1, s2, u)] + ([bias.clone().requires_grad_(True)] if with_bias else [None])
        lb = [t.clone().requires_grad_(True) for t in (x, s1, s2, u)] + ([bias.clone().requires_grad_(True)] if with_bias else [None])
        ya = DiagApplyFunction.apply(*la, S, True, relu_in, relu_out)
        yb = DiagApplyFunction.apply(torch.relu(lb[0]) if relu_in else lb[0], *lb[1:], S, True)
        yb = torch.relu(yb) if relu_out else yb
        assert torch.equal(ya, yb)
        ga = torch.autograd.grad(ya, [t for t in la if t is not None], gout)
        gb = torch.autograd.grad(yb, [t for t in lb if t is not None], gout)
        for name, a, c in zip(("x", "s1", "s2", "u", "bias"), ga, gb):
            assert torch.equal(a, c), (name, with_bias)
    if dtype == torch.float64 and not shared:
        leaves = [t[..., :8].clone().requires_grad_(True) if t.dim() else t for t in (x[:, :5], s1, s2, u, bias)]
        leaves = [x[:, :5, :8].clone().requires_grad_(True), s1[:8].clone().requires_grad_(True), s2[:8].clone().requires_grad_(True),
                  u[:, :8].clone().requires_grad_(True), bias[:, :8].clone().requires_grad_(True)]
        fn = lambda *a: DiagApplyFunction.apply(*a, S, True, relu_in, relu_out)       # noqa: E731
        assert torch.autograd.gradcheck(fn, leaves, eps=1e-6, atol=1e-7)
        assert torch.autograd.gradgradcheck(fn, leaves, eps=1e-6, atol=1e-6)


def test_network_folds_relu_into_the_square_layer(monkeypatch, hip_lib):
    """``WHVINetwork.forward_batched``: an nn.ReLU in front of / behind a square WHVI layer on the GPU is folded into that
    layer's launch (BASELINE config 4's 3 -> D -> D -> 1 network: both of its activations).  Same predictions, loss and
    gradients as with the activations run as passes of their own; the per-sample loop and the faithful dataflow never fuse."""
    import torch.nn as nn
    import whvi_amd.networks as networks
    from whvi_amd.layers import WHVILinear
    from whvi_amd.networks import WHVIRegression
    torch.manual_seed(8)
    net = WHVIRegression([WHVILinear(3, 64), nn.ReLU(), WHVILinear(64, 64, bias=True), nn.ReLU(), WHVILinear(64, 1)],
                         train_samples=5, eval_samples=5)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.rsplit(".", 1)[-1] in ("s1", "s2"):
                p.mul_(30.0)
            if name.endswith("g_mu") or name.endswith("bias"):
                p.normal_()
    net = net.to(DEV).train()
    x, y = torch.randn(40, 3, device=DEV), torch.randn(40, 1, device=DEV)
    calls = []
    real = WHVISquarePow2Matrix.forward_mc
    monkeypatch.setattr(WHVISquarePow2Matrix, "forward_mc",
                        lambda self, x_, n, relu_in=False, relu_out=False: (calls.append((relu_in, relu_out)), real(self, x_, n, relu_in, relu_out))[1])

    def run():
        torch.manual_seed(21)
        net.zero_grad(set_to_none=True)
        loss = net.loss(x, y, n=400)
        loss.backward()
        return loss.detach(), [p.grad.clone() for p in net.parameters()]
    fused_loss, fused_grads = run()
    # both activations are folded: the first into the stacked layer's product (its relu_out), the second into the square layer
    assert calls == [(False, True)], calls
    calls.clear()
    monkeypatch.setattr(networks, "_fuses_relu", lambda module, h: False)
    plain_loss, plain_grads = run()
    assert calls and all(c == (False, False) for c in calls), calls
    assert torch.equal(fused_loss, plain_loss)
    for (name, _), a, b in zip(net.named_parameters(), fused_grads, plain_grads):
        assert torch.equal(a, b), name
