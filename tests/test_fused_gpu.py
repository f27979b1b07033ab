"""GPU parity of the fused scale -> FWHT -> scale -> FWHT -> scale kernel (``whvi_fused_shs_*``)
and of ``WHVILinear`` on the GPU, against the CPU oracle and the reference's recorded bundles.

Bar: bit-exact vs ``oracle.pipeline`` (separate roundings for every multiply, ascending butterfly
order); ``WHVILinear`` forward / KL / backward within 1e-5 relative (BASELINE.json north_star)."""
import math

import numpy as np
import pytest
import torch

import oracle
from oracle import whvi_oracle as wo
from whvi_amd import _hip
from whvi_amd.layers import WHVILinear
from whvi_amd.weights import WBarFunction

from test_host import run_layer_bundle, ReplayRandn, _bundle, _layer_from_bundle, loop_vs_batched

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _bits(a):
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


@pytest.mark.parametrize("log2d", [2, 3, 5, 6, 8, 9, 11, 12, 13])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_col_axis_bit_exact(log2d, dtype, hip_lib):
    if dtype == np.float64 and log2d > 12:
        pytest.skip("f64 rows are limited to 4096")
    d, S, B = 1 << log2d, 3, 5
    rng = np.random.default_rng(log2d)
    x = rng.standard_normal((B * S, d)).astype(dtype)            # (batch, sample, D) row order
    a, c = rng.standard_normal(d).astype(dtype), rng.standard_normal(d).astype(dtype)
    b = rng.standard_normal((S, d)).astype(dtype)
    for use in ((1, 1, 1), (0, 1, 0), (1, 0, 1), (0, 0, 0)):
        aa, bb, cc = (a if use[0] else None), (b if use[1] else None), (c if use[2] else None)
        want = oracle.pipeline(x, aa, bb, cc, n_samples=S, sample_stride=1, axis="col")
        got = _hip.fused_shs(_t(x), _t(aa), _t(bb), _t(cc), axis="col", n_samples=S, sample_stride=1).cpu().numpy()
        assert np.array_equal(_bits(got), _bits(want)), f"log2d={log2d} use={use}"
    # (sample, batch, D) row order
    want = oracle.pipeline(x, a, b, c, n_samples=S, sample_stride=B, axis="col")
    got = _hip.fused_shs(_t(x), _t(a), _t(b), _t(c), axis="col", n_samples=S, sample_stride=B).cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want))


@pytest.mark.parametrize("log2d,dtype", [(4, np.float32), (9, np.float32), (11, np.float32), (12, np.float32), (11, np.float64)])
def test_col_axis_special_values(log2d, dtype, hip_lib):
    """Zeros of both signs, subnormals, huge values, inf and NaN in the data AND in the scale vectors, through the fused
    pipeline (its lane stages run as signed fused multiply-adds): NaN in the same places as the oracle, every other
    value equal, and bit-identical wherever the oracle's value is not a zero."""
    d, S, B = 1 << log2d, 3, 11
    rng = np.random.default_rng(40 + log2d)
    info = np.finfo(dtype)
    specials = np.array([0.0, -0.0, info.tiny, -info.tiny, info.tiny / 4, -info.tiny / 8, info.max, -info.max, info.max / 2,
                         1.0, -1.0, 3.0, np.inf, -np.inf, np.nan], dtype=dtype)
    x = rng.standard_normal((B * S, d)).astype(dtype)
    x[1, :15] = specials
    x[2] = specials[rng.integers(0, 12, d)]                       # finite specials: overflow, cancellation, subnormal sums
    x[3] = specials[rng.integers(0, 6, d)]
    x[4] = -0.0
    x[5] = 0.0
    a, c = rng.standard_normal(d).astype(dtype), rng.standard_normal(d).astype(dtype)
    b = rng.standard_normal((S, d)).astype(dtype)
    a[::5], c[::7], b[1, ::3] = 0.0, -0.0, 0.0                    # zero scales of both signs
    a[1], c[2], b[2, 3] = -1.0, info.tiny, info.max / 4
    with np.errstate(all="ignore"):
        want = oracle.pipeline(x, a, b, c, n_samples=S, sample_stride=1, axis="col")
    got = _hip.fused_shs(_t(x), _t(a), _t(b), _t(c), axis="col", n_samples=S, sample_stride=1).cpu().numpy()
    nan_w, nan_g = np.isnan(want), np.isnan(got)
    assert np.array_equal(nan_w, nan_g)
    ok = ~nan_w
    assert np.array_equal(got[ok], want[ok])
    nz = ok & (want != 0)
    assert np.array_equal(_bits(got)[nz], _bits(want)[nz])


@pytest.mark.parametrize("log2d", [2, 4, 7, 9, 10, 12])
def test_row_axis_and_identity_input_bit_exact(log2d, hip_lib):
    d, S = 1 << log2d, 3
    rng = np.random.default_rng(100 + log2d)
    s1, s2 = rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)
    u = rng.standard_normal((S, d)).astype(np.float32)
    # general input, per-row scalars, groups of D rows per sample
    x = rng.standard_normal((S * d, d)).astype(np.float32)
    want = oracle.pipeline(x, s1, u, s2, n_samples=S, sample_stride=d, group_rows=d, axis="row")
    got = _hip.fused_shs(_t(x), _t(s1), _t(u), _t(s2), axis="row", n_samples=S, sample_stride=d, group_rows=d)
    assert np.array_equal(_bits(got.cpu().numpy()), _bits(want))
    # identity input synthesised in-kernel == explicit torch.diag(s2) (src/weights.py:73)
    eye = np.tile(np.eye(d, dtype=np.float32), (S, 1))
    want = oracle.pipeline(eye, s1, u, s2, n_samples=S, sample_stride=d, group_rows=d, axis="row")
    got = _hip.fused_shs(None, _t(s1), _t(u), _t(s2), axis="row", n_samples=S, sample_stride=d, group_rows=d,
                         rows=S * d, d=d, dtype=torch.float32, device=torch.device(DEV))
    assert np.array_equal(_bits(got.cpu().numpy()), _bits(want))
    # ... which is exactly D * diag(s1 * u * s2) (SURVEY.md finding 1)
    for k in range(S):
        assert np.array_equal(got.cpu().numpy()[k * d:(k + 1) * d],
                              np.diag(s1 * (np.float32(d) * (u[k] * s2))).astype(np.float32))


def test_fused_equals_two_plain_fwht_calls(hip_lib):
    """Fusion changes traffic, not values: same bits as scale / fwht / scale / fwht / scale in torch."""
    d, rows = 2048, 512
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(rows, d, device=DEV, generator=g)
    a, c = torch.randn(d, device=DEV, generator=g), torch.randn(d, device=DEV, generator=g)
    b = torch.randn(64, d, device=DEV, generator=g)
    fused = _hip.fused_shs(x, a, b, c, axis="col", n_samples=64, sample_stride=1)
    bs = b[torch.arange(rows, device=DEV) % 64]
    plain = a * _hip.fwht_rows(bs * _hip.fwht_rows(c * x))
    assert torch.equal(fused, plain)


def test_wbar_function_gradcheck_float64(hip_lib):
    D, S, J = 8, 2, 3
    s1 = torch.randn(J, D, dtype=torch.float64, device=DEV, requires_grad=True)
    s2 = torch.randn(J, D, dtype=torch.float64, device=DEV, requires_grad=True)
    u = torch.randn(J, S, D, dtype=torch.float64, device=DEV, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b, c: WBarFunction.apply(a, b, c, None), (s1, u, s2))
    # first-row-only mode of the column layer
    s1, s2, u = s1[:1].detach().requires_grad_(), s2[:1].detach().requires_grad_(), u[:1, :1].detach().requires_grad_()
    assert torch.autograd.gradcheck(lambda a, b, c: WBarFunction.apply(a, b, c, 1), (s1, u, s2))
    full = WBarFunction.apply(s1, u, s2, None)
    assert torch.equal(WBarFunction.apply(s1, u, s2, 1), full[:, :, :1])


def test_per_sample_outer_scales_bit_exact(hip_lib):
    """whvi_fused_shs_ex with A/C per sample == separate launches per sample (stacked sub-matrices)."""
    d, J = 16, 5
    rng = np.random.default_rng(9)
    s1, s2, u = (rng.standard_normal((J, d)).astype(np.float32) for _ in range(3))
    got = _hip.fused_shs(None, _t(s1), _t(u), _t(s2), axis="row", n_samples=J, sample_stride=d, group_rows=d,
                         rows=J * d, d=d, dtype=torch.float32, device=torch.device(DEV),
                         a_per_sample=True, c_per_sample=True).cpu().numpy()
    for j in range(J):
        assert np.array_equal(got[j * d:(j + 1) * d], wo.w_bar(s1[j], s2[j], u[j]))
    # column axis: every sample with its own a, b, c
    x = rng.standard_normal((J * 3, d)).astype(np.float32)        # (sample, batch 3, D)
    got = _hip.fused_shs(_t(x), _t(s1), _t(u), _t(s2), axis="col", n_samples=J, sample_stride=3,
                         a_per_sample=True, c_per_sample=True).cpu().numpy()
    for j in range(J):
        want = oracle.pipeline(x[3 * j:3 * j + 3], s1[j], u[j][None], s2[j], n_samples=1, axis="col")
        assert np.array_equal(got[3 * j:3 * j + 3], want)


@pytest.mark.parametrize("name", ["sq8", "sq64b", "sq512", "sq4096", "st3x16", "st5x7b", "st13x128",
                                  "col1x10b", "col16x1"])
def test_layer_forward_kl_backward_vs_reference_gpu(name, monkeypatch, hip_lib, dataflow):
    run_layer_bundle(name, DEV, monkeypatch, rtol=1e-5)


def test_square_layer_matches_numpy_oracle_tightly(monkeypatch, hip_lib, dataflow):
    """D = 4096: the reference's host path is butterflies too, so the GPU layer, the oracle and the
    recorded reference output agree to the last bits (softplus ulp aside)."""
    b = _bundle("sq4096")
    layer = _layer_from_bundle(b, DEV)
    monkeypatch.setattr(torch, "randn", ReplayRandn([b["eps0"]]))
    y = layer(torch.from_numpy(b["x"]).to(DEV)).detach().cpu().numpy()
    monkeypatch.undo()
    scale = np.abs(b["y"]).max()
    assert np.abs(y - b["y"]).max() <= 2e-7 * scale
    params = {k[len("param."):]: v for k, v in b.items() if k.startswith("param.")}
    want = wo.layer_from_params(4096, 4096, 0.7, params).forward(b["x"], b["eps0"])
    assert np.abs(y - want).max() <= 2e-7 * scale


def test_network_on_gpu_equals_host_path(monkeypatch, hip_lib, dataflow):
    """A 3 -> 64 -> 64 -> 1 network with all three layer flavours: loss, predictions and every gradient on the GPU
    (fused kernels) against the host path (the reference's op chain under plain autograd) for the same parameters and
    the same eps -- 1e-5 relative on values, 3e-5 on gradients (the host's dense-H matmul carries its own rounding)."""
    import copy
    import torch.nn as nn
    from whvi_amd.networks import WHVIRegression
    from test_host import EpsRouter
    torch.manual_seed(21)
    S = 2
    host = WHVIRegression([WHVILinear(3, 64, bias=True), nn.ReLU(), WHVILinear(64, 64), nn.ReLU(), WHVILinear(64, 1)],
                          train_samples=S, eval_samples=3)
    with torch.no_grad():
        for n_, p_ in host.named_parameters():
            if n_.endswith("g_mu") or n_.endswith("s1") or n_.endswith("s2"):
                p_.copy_(torch.randn(p_.shape) * 0.3)
    dev = copy.deepcopy(host).to(DEV)
    g = torch.Generator().manual_seed(5)
    # one table per layer WIDTH: the stacked layer has 16 sub-matrices of D = 4, square and column layers share D = 64
    # but draw in a fixed order (square first), which EpsRouter's per-D call counter follows
    tables = {4: torch.randn(S, 16, 4, generator=g), 64: torch.randn(2 * S, 1, 64, generator=g)}
    x, y = torch.randn(17, 3, generator=g), torch.randn(17, 1, generator=g)
    out = {}
    real = torch.randn
    for name, net, device in (("host", host, "cpu"), ("gpu", dev, DEV)):
        net.mc_mode = "loop"
        net.train()
        monkeypatch.setattr(torch, "randn", EpsRouter({4: tables[4], 64: tables[64]}, "loop", real))
        loss = net.loss(x.to(device), y.to(device), 100)
        monkeypatch.undo()
        loss.backward()
        out[name] = (float(loss), torch.cat([p.grad.reshape(-1).cpu() for p in net.parameters()]))
    assert abs(out["host"][0] - out["gpu"][0]) <= 1e-5 * abs(out["host"][0])
    scale = float(out["host"][1].abs().max())
    assert float((out["host"][1] - out["gpu"][1]).abs().max()) <= 3e-5 * scale
    dev.eval()
    assert dev(x.to(DEV)).shape == (17, 1, 3)


def test_batched_mc_pass_equals_loop_gpu(monkeypatch, hip_lib, dataflow):
    loop_vs_batched(DEV, monkeypatch)


def test_partial_identity_groups(hip_lib):
    """group_rows < D with the identity input: only the first rows of every matrix are produced."""
    d, J, R = 64, 4, 3
    rng = np.random.default_rng(21)
    s1, s2, u = (rng.standard_normal((J, d)).astype(np.float32) for _ in range(3))
    got = _hip.fused_shs(None, _t(s1[:, :R]), _t(u[:, :R]), _t(s2[:, :R]), axis="row", n_samples=J, sample_stride=R,
                         group_rows=R, rows=J * R, d=d, dtype=torch.float32, device=torch.device(DEV),
                         a_per_sample=True, c_per_sample=True).cpu().numpy()
    for j in range(J):
        assert np.array_equal(got[j * R:(j + 1) * R], wo.w_bar(s1[j], s2[j], u[j])[:R])


def test_graphed_predictor_replays_fresh_samples(hip_lib):
    """hipGraph replay of the batched predictive pass: same distribution as eager, new eps every replay,
    and it tracks parameter updates made in place (the graph reads the live parameter tensors)."""
    import torch.nn as nn
    from whvi_amd.graphs import GraphedPredictor
    from whvi_amd.networks import WHVIRegression
    torch.manual_seed(0)
    net = WHVIRegression([nn.Linear(2, 32), nn.Tanh(), WHVILinear(32, 32), nn.Tanh(), nn.Linear(32, 1, bias=False)],
                         eval_samples=256).to(DEV).eval()
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith("s1") or n_.endswith("s2"):
                p_.mul_(30.0)
            if n_.endswith("g_mu"):
                p_.add_(0.5)
    x = torch.randn(64, 2, device=DEV)
    gp = GraphedPredictor(net, x)
    a = gp(x).clone()
    b = gp(x).clone()
    assert a.shape == (64, 1, 256) and torch.isfinite(a).all()
    assert not torch.equal(a, b), "every replay must draw fresh eps"
    with torch.no_grad():
        eager = net.forward_batched(x, 256)
    # Monte-Carlo means over 256 samples agree within sampling error (a few standard errors)
    se = eager.std(dim=2) / 16 + 1e-6
    assert float(((a.mean(dim=2) - eager.mean(dim=2)).abs() / se).max()) < 8.0
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith("g_mu"):
                p_.zero_()
            if n_.endswith("g_rho"):
                p_.fill_(-30.0)       # sigma ~ 1e-13: W = 0 up to noise -> tanh(0) = 0 -> output 0
    assert float(gp(x).abs().max()) < 1e-6


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rows_shorter_than_a_chunk(dtype, hip_lib):
    """D = 1, 2 (f64: D = 1) take the thread-per-row kernel: every axis / identity / per-sample combination."""
    rng = np.random.default_rng(5)
    for d in ((1, 2) if dtype == np.float32 else (1,)):
        S, B = 3, 7
        x = rng.standard_normal((B * S, d)).astype(dtype)
        a, c = rng.standard_normal(d).astype(dtype), rng.standard_normal(d).astype(dtype)
        b = rng.standard_normal((S, d)).astype(dtype)
        want = oracle.pipeline(x, a, b, c, n_samples=S, sample_stride=1, axis="col")
        got = _hip.fused_shs(_t(x), _t(a), _t(b), _t(c), axis="col", n_samples=S, sample_stride=1).cpu().numpy()
        assert np.array_equal(_bits(got), _bits(want))
        # identity input: what WHVILinear(2, n) / WHVILinear(1, 1) run on the GPU
        s1, s2, u = (rng.standard_normal((S, d)).astype(dtype) for _ in range(3))
        got = _hip.fused_shs(None, _t(s1), _t(u), _t(s2), axis="row", n_samples=S, sample_stride=d, group_rows=d,
                             rows=S * d, d=d, dtype=torch.from_numpy(x).dtype, device=torch.device(DEV),
                             a_per_sample=True, c_per_sample=True).cpu().numpy()
        for k in range(S):
            eye = np.eye(d, dtype=dtype)
            want = oracle.pipeline(eye, s1[k], u[k][None], s2[k], n_samples=1, sample_stride=d, group_rows=d, axis="row")
            assert np.array_equal(_bits(got[k * d:(k + 1) * d]), _bits(want))


def test_tiny_layers_on_gpu(monkeypatch, hip_lib):
    """Shapes whose square blocks have D < 4 -- WHVILinear(2, 5) (stack of 2x2), (2, 2), (1, 1), (1, 3), (3, 1),
    (2, 1): the batched pass (rows shorter than one 16-byte chunk take the thread-per-row kernels) equals one
    stochastic pass per sample with the same eps; values against the oracle are in the next test."""
    for n_in, n_out in ((2, 5), (2, 2), (1, 1), (1, 3), (3, 1), (2, 1)):
        torch.manual_seed(n_in * 7 + n_out)
        layer = WHVILinear(n_in, n_out, bias=True).to(DEV)
        with torch.no_grad():
            for n_, p_ in layer.named_parameters():
                if n_.endswith("g_mu") or n_.endswith("s1") or n_.endswith("s2"):
                    p_.copy_(torch.randn(p_.shape, device=DEV) * 0.5)
        sub = layer.weight_submodule
        J = getattr(sub, "stack", 1)
        D = getattr(sub, "D_in", None) or getattr(sub, "D_adjusted", None) or sub.D
        S = 3
        eps = np.random.default_rng(n_in + 10 * n_out).standard_normal((S, J, D)).astype(np.float32)
        x = torch.randn(6, n_in, device=DEV)
        with torch.no_grad():
            monkeypatch.setattr(torch, "randn", ReplayRandn([np.ascontiguousarray(np.swapaxes(eps, 0, 1))]))
            batched = layer.forward_mc(x, S)
            monkeypatch.undo()
            for k in range(S):
                monkeypatch.setattr(torch, "randn", ReplayRandn([eps[k, j] for j in range(J)]))
                one = layer(x)
                monkeypatch.undo()
                assert one.shape == (6, n_out)
                assert float((batched[k] - one).abs().max()) <= 1e-6 * max(float(one.abs().max()), 1e-20), (n_in, n_out, k)


@pytest.mark.parametrize("n_in,n_out", [(2, 5), (2, 2), (1, 1), (1, 3), (3, 1), (2, 1), (5, 7), (7, 2), (16, 4),
                                        (6, 64), (64, 64), (33, 100)])
def test_layer_values_vs_numpy_oracle_gpu(n_in, n_out, monkeypatch, hip_lib, dataflow):
    """Forward values of every WHVILinear flavour on the GPU against the numpy restatement of the reference
    (oracle/whvi_oracle.py, itself pinned to the reference's recorded bundles), eps replayed; 1e-5 relative."""
    torch.manual_seed(n_in * 131 + n_out)
    layer = WHVILinear(n_in, n_out, lambda_=0.3, bias=True)
    with torch.no_grad():
        for n_, p_ in layer.named_parameters():
            if n_.endswith("g_mu") or n_.endswith("bias"):
                p_.copy_(torch.randn(p_.shape) * 0.4)
            if n_.endswith("s1") or n_.endswith("s2"):
                p_.mul_(20.0)
    params = {k: v.detach().numpy().copy() for k, v in layer.named_parameters()}
    ref = wo.layer_from_params(n_in, n_out, 0.3, params)
    x = torch.randn(9, n_in)
    n_draws = ref.stack if isinstance(ref, wo.Stacked) else 1
    d_eps = ref.D_in if isinstance(ref, wo.Stacked) else (ref.square.D if isinstance(ref, wo.Column) else ref.D)
    eps = [np.random.default_rng(7 + i).standard_normal(d_eps).astype(np.float32) for i in range(n_draws)]
    want = ref.forward(x.numpy(), eps if isinstance(ref, wo.Stacked) else eps[0])
    layer = layer.to(DEV)
    monkeypatch.setattr(torch, "randn", ReplayRandn(eps))
    got = layer(x.to(DEV)).detach().cpu().numpy()
    monkeypatch.undo()
    scale = float(np.abs(want).max()) or 1.0
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-5 * scale
    assert abs(float(layer.kl) - float(ref.kl)) <= 1e-5 * abs(float(ref.kl))


def test_exploit_diagonal_is_bit_identical(monkeypatch, hip_lib):
    """The opt-in diagonal shortcut returns the same bits as the faithful FWHT + GEMM path (finite inputs)."""
    torch.manual_seed(1)
    layer = WHVILinear(256, 256, bias=True).to(DEV)
    with torch.no_grad():
        layer.weight_submodule.g_mu.copy_(torch.randn(256) * 0.3)
    sq = layer.weight_submodule
    h = torch.randn(50, 256, device=DEV)
    eps = [np.random.default_rng(3).standard_normal(256).astype(np.float32)]
    outs = []
    for flag in (False, True):
        sq.exploit_diagonal = flag
        monkeypatch.setattr(torch, "randn", ReplayRandn(eps))
        outs.append(layer(h).detach())
        monkeypatch.undo()
    assert torch.equal(outs[0], outs[1])
    eps4 = [np.random.default_rng(10 + i).standard_normal(256).astype(np.float32) for i in range(4)]
    mc = []
    for flag in (False, True):
        sq.exploit_diagonal = flag
        monkeypatch.setattr(torch, "randn", ReplayRandn(eps4))
        mc.append(layer.forward_mc(h, 4).detach())
        monkeypatch.undo()
    assert torch.allclose(mc[0], mc[1], rtol=1e-6, atol=0)     # batched GEMM may reorder nothing but zeros


def test_reparam_kl_kernel_vs_torch_ops(hip_lib):
    """F3: one launch == softplus / multiply / stack / kl_diag_normal of the reference, forward and backward."""
    import torch.nn.functional as F
    from whvi_amd.utils import kl_diag_normal
    from whvi_amd.weights import ReparamKLFunction
    torch.manual_seed(2)
    J, S, D, lam = 3, 5, 700, 0.37            # D not a multiple of 256: partial last block
    g_mu = (torch.randn(J, D, device=DEV) * 0.5).requires_grad_()
    g_rho = (torch.rand(J, D, device=DEV) * 25 - 4).requires_grad_()     # crosses softplus' threshold of 20
    eps = torch.randn(J, S, D, device=DEV)
    u, kl = ReparamKLFunction.apply(g_mu, g_rho, eps, lam)
    sig = F.softplus(g_rho)
    u_ref = torch.cat((g_mu.unsqueeze(1), sig.unsqueeze(1) * eps), dim=1)
    kl_ref = torch.stack([kl_diag_normal(g_mu[j], sig[j], torch.zeros(D, device=DEV), torch.ones(D, device=DEV) * lam)
                          for j in range(J)])
    assert torch.allclose(u, u_ref, rtol=1e-6, atol=1e-7)
    assert torch.allclose(kl, kl_ref, rtol=1e-5)
    wu, wk = torch.randn_like(u), torch.randn(J, device=DEV)
    ga = torch.autograd.grad((u * wu).sum() + (kl * wk).sum(), (g_mu, g_rho))
    gb = torch.autograd.grad((u_ref * wu).sum() + (kl_ref * wk).sum(), (g_mu, g_rho))
    for a_, b_ in zip(ga, gb):
        assert torch.allclose(a_, b_, rtol=1e-4, atol=1e-5 * float(b_.abs().max()))


def test_kl_property_tolerance_contract(hip_lib):
    """``layer.kl`` on the GPU in float32 is one launch of the reparameterisation + KL kernel: the same D terms as
    ``kl_diag_normal`` (src/utils.py:49-71, the reference's argument convention) in another summation order.  Contract
    (weights._posterior_kl): equal to the formula evaluated as torch ops within 2e-6 relative -- not bit for bit -- for
    every layer flavour; float64 parameters evaluate the formula itself, bit for bit; lambda <= 0 raises."""
    from whvi_amd.utils import kl_diag_normal
    torch.manual_seed(21)
    for n_in, n_out in ((512, 512), (4096, 4096), (3, 1024), (1024, 1), (1, 10)):
        layer = WHVILinear(n_in, n_out, lambda_=0.37).to(DEV)
        want = 0.0
        with torch.no_grad():
            for m in layer.modules():
                if type(m).__name__ in ("WHVISquarePow2Matrix", "_PackedSubMatrix"):
                    m.g_mu.normal_(0.0, 0.5)
                    sd = torch.nn.functional.softplus(m.g_rho).double()
                    want = want + kl_diag_normal(m.g_mu.double(), sd, torch.zeros_like(sd), torch.full_like(sd, 0.37))
        got = layer.kl
        assert got.dtype == torch.float32 and abs(float(got) - float(want)) <= 2e-6 * abs(float(want)), (n_in, n_out)
        f64 = layer.double()
        formula = sum(kl_diag_normal(m.g_mu, torch.nn.functional.softplus(m.g_rho), torch.zeros_like(m.g_mu),
                                     torch.ones_like(m.g_mu) * 0.37)
                      for m in f64.modules() if type(m).__name__ == "WHVISquarePow2Matrix")
        if (n_in, n_out) == (3, 1024):       # stacked: one formula evaluation over all 256 sub-matrices instead of 256 sums
            assert abs(float(f64.kl) - float(formula)) <= 1e-12 * abs(float(formula))
        else:
            assert torch.equal(f64.kl, formula), "float64: the reference's formula itself"
    bad = WHVILinear(8, 8, lambda_=0.0).to(DEV)
    with pytest.raises(RuntimeError, match="lambda must be positive"):
        bad.kl


def test_graphed_train_step_learns(hip_lib):
    """Whole-step hipGraph replay (loss + backward through the fused kernels + Adam): the fit improves and the
    variational parameters move."""
    import torch.nn as nn
    from whvi_amd.graphs import GraphedTrainStep
    from whvi_amd.networks import WHVIRegression
    torch.manual_seed(0)
    net = WHVIRegression([nn.Linear(1, 32), nn.Tanh(), WHVILinear(32, 32, lambda_=1.0), nn.Tanh(), nn.Linear(32, 1)],
                         train_samples=2).to(DEV).train()
    x = torch.linspace(-1, 1, 64, device=DEV).unsqueeze(1)
    y = torch.sin(3 * x)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, capturable=True)
    for _ in range(2):            # the network has already trained eagerly on the default stream ...
        opt.zero_grad(set_to_none=False)
        net.loss(x, y, n=64).backward()
        opt.step()
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    step = GraphedTrainStep(net, opt, x, y, n=64)          # ... and is captured afterwards
    first = float(step(x, y))
    for _ in range(300):
        last = float(step(x, y))
    assert last < first and torch.isfinite(torch.tensor(last))
    moved = [k for k, v in net.named_parameters() if not torch.equal(v.detach(), before[k])]
    assert any(k.endswith("g_rho") for k in moved) and any(k.endswith("s1") for k in moved)
    net.eval()
    with torch.no_grad():
        rmse = float(torch.sqrt(((net(x).mean(dim=2) - y) ** 2).mean()))
    assert rmse < 0.5


def test_graphed_train_step_rejects_a_stale_autograd_graph(hip_lib):
    """A `loss` of an earlier eager pass kept alive (here: a list of losses) leaves the parameters' gradient
    accumulators on the default stream; capturing a backward pass through them aborts inside the HIP runtime.
    GraphedTrainStep must notice during its warm-up and raise -- and work once the tensors are dropped."""
    import torch.nn as nn
    from whvi_amd.graphs import GraphedTrainStep
    from whvi_amd.networks import WHVIRegression
    torch.manual_seed(0)
    net = WHVIRegression([nn.Linear(1, 32), nn.Tanh(), WHVILinear(32, 32, lambda_=1.0), nn.Tanh(), nn.Linear(32, 1)],
                         train_samples=2).to(DEV).train()
    x = torch.linspace(-1, 1, 64, device=DEV).unsqueeze(1)
    y = torch.sin(3 * x)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, capturable=True)
    kept = []
    for _ in range(2):
        opt.zero_grad(set_to_none=False)
        loss = net.loss(x, y, n=64)
        loss.backward(retain_graph=True)
        kept.append(loss)
        opt.step()
    torch.cuda.synchronize()
    params = {k: v.detach().clone() for k, v in net.named_parameters()}
    adam_steps = [float(opt.state[p]["step"]) for p in net.parameters()]
    with pytest.raises(RuntimeError, match="autograd graph of an earlier pass is still alive"):
        GraphedTrainStep(net, opt, x, y, n=64)
    # the error is raised BEFORE the warm-up -- nothing of the training state has moved
    assert all(torch.equal(v.detach(), params[k]) for k, v in net.named_parameters())
    assert [float(opt.state[p]["step"]) for p in net.parameters()] == adam_steps
    kept.clear()
    del loss
    step = GraphedTrainStep(net, opt, x, y, n=64)
    assert torch.isfinite(step(x, y))


@pytest.mark.parametrize("shape", ["toy", "config4"])
def test_building_a_graphed_train_step_has_no_side_effects(shape, hip_lib):
    """the capture warm-up used to leave max(1, warmup) real optimizer steps behind.  Now parameters, Adam's
    moments and step counts, the learning-rate schedule, the device generator and the in-kernel Philox states are put
    back: building the step changes nothing, on a fresh optimizer (whose lazily created state must read as zero) and on
    one that has trained; the first replay then equals the first eager step of an identical twin."""
    import copy
    import torch.nn as nn
    from whvi_amd.evaluation import make_optimizer
    from whvi_amd.graphs import GraphedTrainStep
    from whvi_amd.networks import WHVIRegression
    torch.manual_seed(3)
    if shape == "toy":
        net = WHVIRegression([WHVILinear(3, 16, lambda_=1.0), nn.Tanh(), WHVILinear(16, 16, lambda_=1.0), nn.Tanh(),
                              WHVILinear(16, 1, lambda_=1.0)], train_samples=2).to(DEV).train()
        x, y = torch.randn(32, 3, device=DEV), torch.randn(32, 1, device=DEV)
    else:       # BASELINE config 4's network as bench.py's extras.config4_train_step times it: batch 256, packed parameters
        net = WHVIRegression([WHVILinear(3, 1024), nn.ReLU(), WHVILinear(1024, 1024), nn.ReLU(), WHVILinear(1024, 1)],
                             train_samples=1).to(DEV).train()
        x = torch.randn(256, 3, device=DEV)
        y = torch.sin(x.sum(dim=1, keepdim=True))
    twin = copy.deepcopy(net)
    packed = shape == "config4"
    opt, sched = make_optimizer(net, lambda0=0.05, capturable=True, packed=packed)
    opt2, sched2 = make_optimizer(twin, lambda0=0.05, capturable=True, packed=packed)
    assert len(list(net.parameters())) == (13 if packed else 25)
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    rng = torch.cuda.get_rng_state(torch.device(DEV))
    step = GraphedTrainStep(net, opt, x, y, n=320, scheduler=sched, warmup=3)
    assert all(torch.equal(v.detach(), before[k]) for k, v in net.named_parameters())
    assert float(sched.t) == 0.0 and abs(sched.get_last_lr()[0] - 0.05 * 0.05) < 1e-9
    assert torch.equal(torch.cuda.get_rng_state(torch.device(DEV)), rng)
    for p in net.parameters():
        st = opt.state[p]
        assert float(st["step"]) == 0.0 and float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0
    # first replay == first eager step of the twin (same parameters, same generator state -> same eps)
    torch.cuda.set_rng_state(rng, torch.device(DEV))
    loss_graph = float(step(x, y))
    torch.cuda.set_rng_state(rng, torch.device(DEV))
    opt2.zero_grad(set_to_none=True)
    loss_eager = twin.loss(x, y, n=320)
    loss_eager.backward()
    opt2.step()
    sched2.step()
    assert abs(loss_graph - float(loss_eager)) <= 1e-6 * abs(float(loss_eager))
    for (k, a), (_, b) in zip(net.named_parameters(), twin.named_parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7 if shape == "toy" else 2e-6), k
    assert float(sched.t) == 1.0 and abs(sched.get_last_lr()[0] - sched2.get_last_lr()[0]) < 1e-12
    # a host-side schedule is refused, with the remedy in the message
    host = torch.optim.lr_scheduler.LambdaLR(torch.optim.Adam(twin.parameters(), lr=1e-3, capturable=True), lambda t: 1.0)
    del loss_eager
    with pytest.raises(RuntimeError, match="DeviceLambdaLR"):
        GraphedTrainStep(twin, opt2, x, y, n=320, scheduler=host)


def test_reparam_kl_double_backward_vs_torch_ops(hip_lib):
    """Second derivatives through ReparamKLFunction (a gradient penalty on g_rho / g_mu): the create_graph branch must
    differentiate through sigma = softplus(g_rho) -- in 1 / sigma of the KL term and in grad_eps = grad_u * sigma."""
    from whvi_amd.weights import ReparamKLFunction
    J, S, D, lam = 2, 3, 40, 0.7
    g = torch.Generator().manual_seed(5)
    mu0, rho0 = torch.randn(J, D, generator=g) * 0.3, torch.rand(J, D, generator=g) - 3
    eps0, w = torch.randn(J, S, D, generator=g), torch.randn(J, S + 1, D, generator=g).to(DEV)

    def penalty(use_kernel):
        mu, rho, eps = (t.clone().to(DEV).requires_grad_(True) for t in (mu0, rho0, eps0))
        if use_kernel:
            u, kl = ReparamKLFunction.apply(mu, rho, eps, lam)
        else:
            sigma = torch.nn.functional.softplus(rho)
            u = torch.cat((mu.unsqueeze(1), sigma.unsqueeze(1) * eps), dim=1)
            kl = 0.5 * (math.log(lam) * D - torch.log(sigma).sum(1) - D + (sigma / lam).sum(1) + (mu * (mu / lam)).sum(1))
        loss = (u * w).sum() + (u.square() * w).sum() + 3.0 * kl.sum()
        g_mu, g_rho, g_eps = torch.autograd.grad(loss, (mu, rho, eps), create_graph=True)
        (g_mu.square().sum() + g_rho.square().sum() + g_eps.square().sum()).backward()
        return [t.grad.detach().cpu() for t in (mu, rho, eps)]
    for got, want, name in zip(penalty(True), penalty(False), ("mu", "rho", "eps")):
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()), name


@pytest.mark.parametrize("dtype,J,S,D,R", [
    (torch.float32, 1, 1, 4, 4), (torch.float32, 1, 3, 8, 8), (torch.float32, 2, 2, 64, 64), (torch.float32, 3, 2, 16, 5),
    (torch.float32, 1, 2, 256, 256), (torch.float32, 2, 3, 512, 512), (torch.float32, 1, 2, 1024, 1000),
    (torch.float32, 1, 1, 4096, 4096), (torch.float32, 1, 2, 8192, 24), (torch.float32, 5, 1, 32, 1),
    (torch.float64, 1, 2, 2, 2), (torch.float64, 2, 2, 4, 3), (torch.float64, 1, 2, 128, 128), (torch.float64, 1, 1, 4096, 9)])
def test_wbar_backward_kernel_vs_composed_chain(dtype, J, S, D, R, hip_lib):
    """whvi_wbar_bwd (one launch) against the differentiable op chain (4 FWHT launches + elementwise ops, the
    path create_graph=True takes) and, in exact arithmetic, the closed form the as-written matrix implies:
    W = D diag(s1 u s2)  =>  dL/du_i = D s1_i s2_i gW_ii, and likewise for s1, s2 (SURVEY.md finding 1)."""
    g = torch.Generator().manual_seed(J * 1000 + D + R)
    s1, s2 = (torch.randn(J, D, generator=g, dtype=dtype).to(DEV).requires_grad_() for _ in range(2))
    u = torch.randn(J, S, D, generator=g, dtype=dtype).to(DEV).requires_grad_()
    gw = torch.randn(J, S, R, D, generator=g, dtype=dtype).to(DEV)
    rows = None if R == D else R

    W = WBarFunction.apply(s1, u, s2, rows)
    fused = torch.autograd.grad(W, (s1, u, s2), gw)                              # grad mode off in backward -> kernel
    W = WBarFunction.apply(s1, u, s2, rows)
    chain = torch.autograd.grad(W, (s1, u, s2), gw, create_graph=True)           # differentiable chain
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    for name, a, b in zip(("s1", "u", "s2"), fused, chain):
        scale = float(b.abs().max()) or 1.0
        assert a.shape == b.shape
        assert float((a - b.detach()).abs().max()) <= tol * scale * math.sqrt(D), name
    # closed form (float64 on the host)
    d64 = lambda t: t.detach().double().cpu()  # noqa: E731
    diag = torch.diagonal(d64(gw), dim1=2, dim2=3)                               # (J, S, R)
    s1r, s2r, ur = d64(s1)[:, None, :R], d64(s2)[:, None, :R], d64(u)[:, :, :R]
    want = {"u": D * s1r * s2r * diag, "s1": (D * ur * s2r * diag).sum(1), "s2": (D * s1r * ur * diag).sum(1)}
    for name, a in zip(("s1", "u", "s2"), fused):
        w = want[name]
        got = d64(a)[..., :R]
        assert float(d64(a)[..., R:].abs().max() if R < D else 0.0) == 0.0
        # the chain's rounding noise is relative to the FWHT output magnitude (~ sqrt(D) |gw|), not to the diagonal
        noise = (1e-6 if dtype == torch.float32 else 1e-14) * D * float(d64(gw).abs().max()) * 8
        assert float((got - w).abs().max()) <= noise * float(max(s1r.abs().max(), 1) * max(s2r.abs().max(), 1)
                                                             * max(ur.abs().max(), 1)), name
    # the LDS-staged butterfly network (f32 rows of 256 .. 4096, cache-resident sizes) and the DPP network in its signed
    # form (streams) make the same adds in the same order up to exact sign flips, and everything around them is
    # shared: identical values, with and without the mean term
    for mean in (False, True):
        uu = torch.cat((u[:, :1], u), dim=1) if mean else u
        a = _hip.wbar_bwd(gw, s1.detach(), uu.detach(), s2.detach(), mean=mean)
        b = _hip.wbar_bwd(gw, s1.detach(), uu.detach(), s2.detach(), mean=mean, no_lds=True)
        first = 1 if mean else 0
        assert torch.equal(a[:, :, first:, :R], b[:, :, first:, :R]), f"LDS vs DPP network, mean={mean}"


def test_wbar_backward_kernel_rejects_bad_shapes(hip_lib):
    gw = torch.zeros(1, 1, 4, 4, device=DEV)
    with pytest.raises(RuntimeError, match="shapes"):
        _hip.wbar_bwd(gw, torch.zeros(1, 8, device=DEV), torch.zeros(1, 1, 4, device=DEV), torch.zeros(1, 4, device=DEV))
    with pytest.raises(RuntimeError, match="dtypes"):
        _hip.wbar_bwd(gw, torch.zeros(1, 4, device=DEV).double(), torch.zeros(1, 1, 4, device=DEV),
                      torch.zeros(1, 4, device=DEV))
    out = _hip.wbar_bwd(torch.ones(1, 2, 3, 8, device=DEV), torch.ones(1, 8, device=DEV), torch.ones(1, 2, 8, device=DEV),
                        torch.ones(1, 8, device=DEV))
    assert out.shape == (3, 1, 2, 8) and float(out[..., 3:].abs().max()) == 0.0     # rows i >= R: zero gradient
    assert not _hip.wbar_bwd_supported(torch.float32, 2) and not _hip.wbar_bwd_supported(torch.float64, 8192)
    assert _hip.wbar_bwd_supported(torch.float32, 8192) and _hip.wbar_bwd_supported(torch.float64, 2)


def test_reparam_kl_backward_kernel_vs_torch_ops(hip_lib):
    """whvi_reparam_kl_bwd (one launch) against autograd over the reference's op chain (softplus, mul, kl_diag_normal)."""
    from whvi_amd.weights import ReparamKLFunction
    from whvi_amd.utils import kl_diag_normal
    import torch.nn.functional as F
    for J, S, D, lam in ((1, 1, 128, 1e-5), (3, 5, 64, 0.3), (2, 0, 16, 1.0), (1, 7, 1000, 2.0)):
        g = torch.Generator().manual_seed(D + S)
        mu = torch.randn(J, D, generator=g).to(DEV).requires_grad_()
        rho = (torch.rand(J, D, generator=g) * 4 - 3)
        rho[0, 0] = 25.0                                          # above softplus' threshold
        rho = rho.to(DEV).requires_grad_()
        eps = torch.randn(J, S, D, generator=g).to(DEV)
        gu = torch.randn(J, S + 1, D, generator=g).to(DEV)
        gk = torch.randn(J, generator=g).to(DEV)
        u, kl = ReparamKLFunction.apply(mu, rho, eps, lam)
        got = torch.autograd.grad((u, kl), (mu, rho), (gu, gk))
        sigma = F.softplus(rho)
        u2 = torch.cat((mu.unsqueeze(1), sigma.unsqueeze(1) * eps), dim=1)
        kl2 = torch.stack([kl_diag_normal(mu[j], sigma[j], torch.zeros(D, device=DEV), torch.ones(D, device=DEV) * lam)
                           for j in range(J)])
        want = torch.autograd.grad((u2, kl2), (mu, rho), (gu, gk), retain_graph=True)
        for a, b in zip(got, want):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
        # KL-only gradient (the incoming gradient of u is None)
        u, kl = ReparamKLFunction.apply(mu, rho, eps, lam)
        only_kl = torch.autograd.grad(kl.sum(), (mu, rho))
        want_kl = torch.autograd.grad(kl2.sum(), (mu, rho))
        for a, b in zip(only_kl, want_kl):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("m,n_out,n_mc,layout", [(100, 1, 1, "perm"), (37, 3, 5, "perm"), (64, 2, 8, "contig"),
                                                 (1, 1, 1, "contig"), (5000, 1, 64, "perm"), (9, 4, 3, "expand")])
def test_gauss_mnll_kernel_vs_reference_formula(m, n_out, n_mc, layout, hip_lib):
    """whvi_gauss_mnll (one reduction, strided y_hat) against the reference's estimator (src/likelihoods.py:18-29,
    per-output loop restated in float64) and its autograd gradients."""
    from whvi_amd.likelihoods import GaussianLikelihood
    g = torch.Generator().manual_seed(m + n_mc)
    y = torch.randn(m, n_out, generator=g).to(DEV)
    if layout == "perm":                         # what forward_batched returns: (S, batch, out) permuted
        base = torch.randn(n_mc, m, n_out, generator=g).to(DEV).requires_grad_()
    elif layout == "expand":                     # a network without stochastic layers
        base = torch.randn(m, n_out, generator=g).to(DEV).requires_grad_()
    else:
        base = torch.randn(m, n_out, n_mc, generator=g).to(DEV).requires_grad_()
    view = {"perm": lambda t: t.permute(1, 2, 0), "expand": lambda t: t.unsqueeze(2).expand(m, n_out, n_mc),
            "contig": lambda t: t}[layout]
    lik = GaussianLikelihood(0.7).to(DEV)
    n = 4321
    got = lik.mnll_batch_estimate(y, view(base), n)
    g_base, g_sigma = torch.autograd.grad(got, (base, lik.sigma))
    # float64 restatement of the reference loop
    b64 = base.detach().double().requires_grad_()
    yh64 = view(b64)
    s64 = lik.sigma.detach().double().requires_grad_()
    total = 0.0
    for o in range(n_out):
        dist = torch.distributions.Normal(yh64[:, o, :], s64)
        total = total + dist.log_prob(y.double()[:, o].reshape(m, 1)).sum()
    want = -n / (m * n_mc) * total
    w_base, w_sigma = torch.autograd.grad(want, (b64, s64))
    assert abs(float(got) - float(want)) <= 2e-5 * abs(float(want))
    assert float((g_base.double() - w_base).abs().max()) <= 2e-5 * float(w_base.abs().max())
    assert abs(float(g_sigma) - float(w_sigma)) <= 5e-5 * abs(float(w_sigma))


@pytest.mark.parametrize("n_in,n_out", [(8, 8), (64, 64), (512, 512), (3, 16), (5, 7), (13, 128), (100, 33), (4, 1024),
                                        (1, 10), (1, 128), (16, 1), (100, 1), (2, 2), (2, 5), (1, 1)])
def test_layer_gradients_gpu_vs_host_path(n_in, n_out, monkeypatch, hip_lib, dataflow):
    """Every WHVILinear flavour (square / stacked / column / transposed column): forward output and ALL gradients
    (s1, s2, g_mu, g_rho, bias, input) on the GPU -- fused weight kernel, one-launch backward kernels -- against the
    host path, which runs the reference's op chain under plain autograd (src/weights.py:34-41,66-93), with the
    same parameters and the same eps.  North-star tolerance: 1e-5 relative (3e-5 on gradients, whose host
    values carry the dense-H matmul's own rounding)."""
    import copy
    torch.manual_seed(n_in * 1009 + n_out)
    host = WHVILinear(n_in, n_out, lambda_=0.5, bias=True)
    with torch.no_grad():
        for name, p in host.named_parameters():
            if name.endswith("g_mu") or name.endswith("bias"):
                p.copy_(torch.randn(p.shape) * 0.3)
            if name.endswith("s1") or name.endswith("s2"):
                p.mul_(30.0)
    dev = copy.deepcopy(host).to(DEV)
    sub = host.weight_submodule
    if hasattr(sub, "weight_matrices"):
        n_draws, d_eps = sub.stack, sub.D_in
    elif hasattr(sub, "weight_submodule"):
        n_draws, d_eps = 1, sub.weight_submodule.D
    else:
        n_draws, d_eps = 1, sub.D
    eps = [np.random.default_rng(31 + i).standard_normal(d_eps).astype(np.float32) for i in range(n_draws)]
    x = torch.randn(11, n_in)
    w = torch.randn(11, n_out)
    results = []
    for layer, device in ((host, "cpu"), (dev, DEV)):
        xin = x.detach().clone().to(device).requires_grad_()
        monkeypatch.setattr(torch, "randn", ReplayRandn(eps))
        out = layer(xin)
        monkeypatch.undo()
        ((out * w.to(device)).sum() + layer.kl).backward()
        grads = {k: p.grad.detach().cpu() for k, p in layer.named_parameters()}
        grads["input"] = xin.grad.detach().cpu()
        results.append((out.detach().cpu(), grads))
    (out_h, g_h), (out_d, g_d) = results
    assert float((out_h - out_d).abs().max()) <= 1e-5 * max(float(out_h.abs().max()), 1e-30)
    for k in g_h:
        scale = max(float(g_h[k].abs().max()), 1e-30)
        assert float((g_h[k] - g_d[k]).abs().max()) <= 3e-5 * scale, k


@pytest.mark.parametrize("dtype,J,S,D,R", [
    (torch.float32, 1, 1, 4, 4), (torch.float32, 1, 3, 8, 8), (torch.float32, 3, 2, 16, 5), (torch.float32, 2, 2, 64, 64),
    (torch.float32, 1, 5, 128, 128), (torch.float32, 2, 3, 512, 512), (torch.float32, 1, 2, 1024, 1000),
    (torch.float32, 1, 2, 4096, 4096), (torch.float32, 1, 2, 8192, 24), (torch.float32, 256, 2, 4, 4), (torch.float32, 5, 1, 32, 1),
    (torch.float64, 1, 2, 2, 2), (torch.float64, 2, 2, 4, 3), (torch.float64, 1, 2, 128, 128), (torch.float64, 1, 1, 4096, 9)])
def test_wbar_forward_kernel_bit_identical_to_generic_fused_launch(dtype, J, S, D, R, hip_lib):
    """whvi_wbar_fwd (first transform generated from bit parities, optional mean-matrix add) against the generic
    fused launch with the identity input (two real transforms) and a separate torch add: bit for bit, and against
    the CPU oracle's pipeline on the first matrix."""
    g = torch.Generator().manual_seed(J * 77 + D + R)
    s1, s2 = (torch.randn(J, D, generator=g, dtype=dtype).to(DEV) for _ in range(2))
    u = torch.randn(J, S, D, generator=g, dtype=dtype).to(DEV)
    got = _hip.wbar_fwd(s1, u, s2, R)
    ref = _hip.fused_shs(None, a=s1[:, :R].repeat_interleave(S, dim=0), b=u[:, :, :R], c=s2[:, :R].repeat_interleave(S, dim=0),
                         axis="row", n_samples=J * S, sample_stride=R, group_rows=R, rows=J * S * R, d=D, dtype=dtype,
                         device=torch.device(DEV), a_per_sample=True, c_per_sample=True).view(J, S, R, D)
    assert got.shape == (J, S, R, D)
    assert torch.equal(got.view(torch.uint8), ref.view(torch.uint8))
    npdt = np.float32 if dtype == torch.float32 else np.float64
    eye = np.eye(D, dtype=npdt)[:R]
    want = oracle.pipeline(eye, s1[0, :R].cpu().numpy(), u[0, :1, :R].cpu().numpy(), s2[0, :R].cpu().numpy(),
                           n_samples=1, sample_stride=R, group_rows=R, axis="row")
    assert np.array_equal(_bits(got[0, 0].cpu().numpy()), _bits(want))
    # mean-matrix add in the epilogue == a separate add of separately rounded matrices
    base = torch.randn(J, R, D, generator=g, dtype=dtype).to(DEV)
    with_base = _hip.wbar_fwd(s1, u, s2, R, base=base)
    assert torch.equal(with_base.view(torch.uint8), (base.unsqueeze(1) + got).view(torch.uint8))


@pytest.mark.parametrize("dtype,J,S,D,R", [
    (torch.float32, 1, 1, 8, 8), (torch.float32, 2, 3, 64, 64), (torch.float32, 3, 2, 16, 5), (torch.float32, 1, 4, 512, 512),
    (torch.float32, 1, 1, 4096, 100), (torch.float32, 1, 2, 2, 2), (torch.float64, 2, 2, 128, 128), (torch.float64, 1, 3, 4, 3)])
def test_wbar_mean_plus_forward_and_backward(dtype, J, S, D, R, hip_lib):
    """``mean_plus``: W[j,k] = w_bar(u[j,0]) + w_bar(u[j,1+k]) from two launches, bit-identical to building all
    1 + S matrices and adding; its one-launch backward (WHVI_WBAR_MEAN) against the differentiable op chain."""
    g = torch.Generator().manual_seed(J * 31 + D + R)
    s1, s2 = (torch.randn(J, D, generator=g, dtype=dtype).to(DEV).requires_grad_() for _ in range(2))
    u = torch.randn(J, S + 1, D, generator=g, dtype=dtype).to(DEV).requires_grad_()
    gw = torch.randn(J, S, R, D, generator=g, dtype=dtype).to(DEV)
    rows = None if R == D else R
    W = WBarFunction.apply(s1, u, s2, rows, True)
    every = WBarFunction.apply(s1, u, s2, rows)
    assert W.shape == (J, S, R, D)
    assert torch.equal(W.view(torch.uint8), (every[:, :1] + every[:, 1:]).view(torch.uint8))
    fused = torch.autograd.grad(W, (s1, u, s2), gw)
    W = WBarFunction.apply(s1, u, s2, rows, True)
    chain = torch.autograd.grad(W, (s1, u, s2), gw, create_graph=True)
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    for name, a, b in zip(("s1", "u", "s2"), fused, chain):
        b = b.detach()
        assert a.shape == b.shape, name
        assert float((a - b).abs().max()) <= tol * (float(b.abs().max()) or 1.0) * math.sqrt(D) * (S + 1), name


def test_inkernel_philox_reparameterisation(hip_lib):
    """whvi_reparam_kl_philox_f32: the eps it reports reproduce u / sigma / KL bit for bit through the
    injectable-eps kernel; the draw is standard normal (moments, tail mass, no duplicates across the index space),
    repeatable from (seed, offset) and advanced by the kernel itself."""
    J, S, D, lam = 3, 21, 1000, 0.7
    g = torch.Generator().manual_seed(5)
    mu = torch.randn(J, D, generator=g).to(DEV)
    rho = (torch.rand(J, D, generator=g) * 3 - 3).to(DEV)
    state = _hip.new_rng_state(torch.device(DEV), seed=1234)
    u, sigma, kl, eps = _hip.reparam_kl_philox(mu, rho, S, lam, state)
    assert state.tolist() == [1234, 2, 0]                                  # two Philox calls per thread, scratch reset
    u2, sigma2, kl2 = _hip.reparam_kl(mu, rho, eps, lam)
    assert torch.equal(u, u2) and torch.equal(sigma, sigma2) and torch.equal(kl, kl2)
    # repeatable from the state, different after it advanced
    again = _hip.reparam_kl_philox(mu, rho, S, lam, torch.tensor([1234, 0, 0], dtype=torch.int64, device=DEV))[3]
    nxt = _hip.reparam_kl_philox(mu, rho, S, lam, state)[3]
    assert torch.equal(again, eps) and not torch.equal(nxt, eps) and state.tolist() == [1234, 4, 0]
    other_seed = _hip.reparam_kl_philox(mu, rho, S, lam, _hip.new_rng_state(torch.device(DEV), seed=1235))[3]
    assert not torch.equal(other_seed, eps)
    # distribution: a large draw
    big = _hip.reparam_kl_philox(torch.zeros(8, 4096, device=DEV), torch.zeros(8, 4096, device=DEV), 64, 1.0,
                                 _hip.new_rng_state(torch.device(DEV), seed=7))[3].double().flatten()
    n = big.numel()                                                         # 2.1 M samples
    assert torch.isfinite(big).all()
    assert abs(float(big.mean())) < 4 / math.sqrt(n) and abs(float(big.var()) - 1.0) < 6 * math.sqrt(2 / n)
    assert abs(float((big ** 3).mean())) < 0.02 and abs(float((big ** 4).mean()) - 3.0) < 0.05
    for z, p in ((1.0, 0.682689), (2.0, 0.954500), (3.0, 0.997300)):
        assert abs(float((big.abs() < z).double().mean()) - p) < 5 * math.sqrt(p * (1 - p) / n)
    assert torch.unique(big).numel() > 0.95 * n                             # no repeated blocks of the counter space
    grid = big.view(8, 64, 4096)
    for a, b in ((grid[:, :, :-1], grid[:, :, 1:]), (grid[:, :-1], grid[:, 1:]), (grid[:-1], grid[1:])):
        assert abs(float((a * b).mean())) < 5 / math.sqrt(a.numel())        # neighbours along every axis uncorrelated


def test_inkernel_rng_in_the_network_and_under_hipgraph(hip_lib):
    """Opt-in ``set_inkernel_rng``: a training step differentiates through the in-kernel draw (same one-launch
    backward), predictions vary from call to call, and a captured predictive pass draws fresh eps on every replay."""
    import torch.nn as nn
    from whvi_amd.graphs import GraphedPredictor
    from whvi_amd.networks import WHVIRegression
    torch.manual_seed(0)
    net = WHVIRegression([WHVILinear(1, 32), nn.Tanh(), WHVILinear(32, 32), nn.Tanh(), WHVILinear(32, 1)],
                         train_samples=4, eval_samples=8).to(DEV).set_inkernel_rng(True)
    x = torch.linspace(-1, 1, 50, device=DEV).unsqueeze(1)
    y = torch.sin(3 * x)
    net.train()
    loss = net.loss(x, y, n=50)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    assert any(float(p.grad.abs().max()) > 0 for n_, p in net.named_parameters() if n_.endswith("g_rho"))
    net.eval()
    with torch.no_grad():
        a, b = net(x), net(x)
    assert a.shape == (50, 1, 8) and not torch.equal(a, b)
    gp = GraphedPredictor(net, x, 8)
    r1 = gp(x).clone()
    r2 = gp(x).clone()
    assert not torch.equal(r1, r2) and torch.isfinite(r1).all() and torch.isfinite(r2).all()
    # every layer's generator moved on
    states = [m._rng_state for m in net.modules() if getattr(m, "_rng_state", None) is not None]
    assert len(states) == 3 and all(int(s[1]) > 0 and int(s[2]) == 0 for s in states)


def test_packed_stacked_layer_on_gpu(monkeypatch, hip_lib):
    """``pack_parameters()`` on the GPU: same outputs and gradients as the reference layout for the same eps
    (loop and batched paths), checkpoint keys unchanged."""
    import copy
    torch.manual_seed(11)
    plain = WHVILinear(13, 128, lambda_=0.5, bias=True)
    with torch.no_grad():
        for name, p in plain.named_parameters():
            if name.endswith("g_mu") or name.endswith("s1") or name.endswith("s2"):
                p.copy_(torch.randn(p.shape) * 0.4)
    packed = copy.deepcopy(plain)
    packed.weight_submodule.pack_parameters()
    plain, packed = plain.to(DEV), packed.to(DEV)
    sub = packed.weight_submodule
    assert set(packed.state_dict()) == set(plain.state_dict()) and len(list(packed.parameters())) == 5
    assert sub.packed_s1.device.type == "cuda" and sub.weight_matrices[3].s1.device.type == "cuda"
    eps = [np.random.default_rng(40 + i).standard_normal(sub.D_in).astype(np.float32) for i in range(sub.stack)]
    x = torch.randn(9, 13, device=DEV)
    for use_mc in (False, True):
        res = []
        for layer in (plain, packed):
            layer.zero_grad()
            monkeypatch.setattr(torch, "randn", ReplayRandn(eps))
            y = layer.weight_submodule.forward_mc(x, 1)[0] if use_mc else layer(x)
            monkeypatch.undo()
            (y.square().sum() + layer.kl).backward()
            res.append(y.detach())
        assert torch.equal(res[0], res[1])
        for name in ("s1", "s2", "g_mu", "g_rho"):
            want = torch.stack([getattr(m, name).grad for m in plain.weight_submodule.weight_matrices])
            got = getattr(sub, "packed_" + name).grad
            assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()), (use_mc, name)


def test_network_and_likelihood_vs_reference_on_gpu(monkeypatch, hip_lib, dataflow):
    """The reference's recorded WHVIRegression run (tests/golden/network_golden.npz: predictions, MNLL, KL for replayed
    eps) and its likelihood unit tests (test/likelihoods.py:8-56) with the network on the GPU: fused weight kernels,
    one-launch Gaussian MNLL reduction.  1e-5 relative."""
    import os
    import torch.nn as nn
    from whvi_amd.likelihoods import GaussianLikelihood
    from whvi_amd.networks import WHVIRegression
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "network_golden.npz"))
    net = WHVIRegression([nn.Linear(1, 8), nn.ReLU(), WHVILinear(8, 8), nn.ReLU(), nn.Linear(8, 2)],
                         train_samples=3, eval_samples=4)
    net.load_state_dict({k[len("state."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state.")})
    net = net.to(DEV).train()
    net.mc_mode = "loop"                      # the reference's RNG order: one draw per sample
    monkeypatch.setattr(torch, "randn", ReplayRandn([g[f"eps{i}"] for i in range(int(g["n_eps"]))]))
    pred = net(torch.from_numpy(g["x"]).to(DEV))
    monkeypatch.undo()
    want = torch.from_numpy(g["pred"])
    assert float((pred.cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    mnll = net.likelihood.mnll_batch_estimate(torch.from_numpy(g["y"]).to(DEV), pred, 100)
    assert abs(float(mnll) - float(g["mnll"])) <= 1e-5 * abs(float(g["mnll"]))
    assert abs(float(net.kl) - float(g["kl"])) <= 1e-5 * abs(float(g["kl"]))
    mnll.backward()
    assert net.likelihood.sigma.grad is not None and torch.isfinite(net.likelihood.sigma.grad)
    # test/likelihoods.py: the explicit double loop, delta 1e-4
    y = torch.reshape(torch.tensor([0., 1., 2., -1.]), (-1, 1)).to(DEV)
    y_hat = torch.tensor([[0.2, 1.1, 2.2, -1.3], [-0.1, 1.05, 2, -1.1]]).T.unsqueeze(1).to(DEV)
    got = float(GaussianLikelihood(sigma=1.0).to(DEV).mnll_batch_estimate(y, y_hat, 12))
    assert abs(got - float(g["lik_value"])) < 1e-4
    n, m, n_mc, sigma = 116, 24, 80, 15.21
    gen = torch.Generator().manual_seed(1)
    y, y_hat = torch.randn((m, 1), generator=gen), torch.randn((m, 1, n_mc), generator=gen)
    target = 0.0
    for j in range(m):
        tmp = 0.0
        for i in range(n_mc):
            tmp += -(np.log(1 / (np.sqrt(2 * np.pi) * sigma)) - 0.5 * (float(y[j] - y_hat[j, 0, i]) / sigma) ** 2)
        target += tmp / n_mc
    target *= n / m
    got = float(GaussianLikelihood(sigma=sigma).to(DEV).mnll_batch_estimate(y.to(DEV), y_hat.to(DEV), n))
    assert abs(target - got) < 1e-4 * max(1.0, abs(target))


@pytest.mark.parametrize("dtype,J,S,D,R", [(torch.float32, 1, 32, 512, 512), (torch.float32, 256, 16, 4, 4), (torch.float32, 1, 3, 1024, 1024),
                                           (torch.float32, 1, 2, 4096, 4096), (torch.float32, 3, 2, 64, 5), (torch.float64, 2, 3, 256, 256),
                                           (torch.float64, 1, 2, 2048, 7), (torch.float32, 1, 5, 8, 1)])
def test_one_launch_mean_plus_sample_weights_equal_the_two_launch_form(dtype, J, S, D, R, hip_lib):
    """``whvi_wbar_fwd_mean`` (both terms of ``w_bar(g_mu) + w_bar(g_sigma eps_k)``, src/weights.py:93, by the same wave)
    against the two-launch form (mean matrix, then samples with the mean added in the epilogue) and, through it, the
    oracle-pinned generic fused launch: bit for bit, BASELINE config 2's and config 4's layer shapes included; rows of
    one 128-register tile keep the two-launch form."""
    g = torch.Generator().manual_seed(D + S)
    s1, s2 = (torch.randn(J, D, generator=g, dtype=dtype).to(DEV) for _ in range(2))
    u = torch.randn(J, 1 + S, D, generator=g, dtype=dtype).to(DEV)
    one = _hip.wbar_fwd_mean(s1, u, s2, R, inline=True)
    assert _hip.last_kernel().endswith(", false, true>"), _hip.last_kernel()
    two = _hip.wbar_fwd_mean(s1, u, s2, R, inline=False)
    assert not _hip.last_kernel().endswith(", true, true>") and one.shape == (J, S, R, D)
    assert torch.equal(one.view(torch.uint8), two.view(torch.uint8))
    every = _hip.wbar_fwd(s1, u, s2, R)                                        # (J, 1 + S, R, D): the terms on their own
    assert torch.equal(one, every[:, :1] + every[:, 1:])
    W = WBarFunction.apply(s1, u, s2, None if R == D else R, True)             # what the layers call: picks by size
    assert torch.equal(W, one)
    long_d = 8192 if dtype == torch.float32 else 4096
    with pytest.raises(RuntimeError, match="two-launch form only"):
        _hip.wbar_fwd_mean(torch.zeros(1, long_d, dtype=dtype, device=DEV), torch.zeros(1, 2, long_d, dtype=dtype, device=DEV),
                           torch.zeros(1, long_d, dtype=dtype, device=DEV), 1, inline=True)


@pytest.mark.parametrize("key", ["f32_D64", "f32_D512", "f32_D2048", "f64_D64", "f64_D512"])
def test_column_pipeline_vs_vectors_recorded_from_the_reference_gpu(key, hip_lib):
    """``whvi_fused_shs_*`` (axis = COL) straight against vectors the LIVE reference composed from its own
    ``matmul_diag_right`` and FWHT function (tests/golden/pipeline_golden.npz, make_golden_r3.py) -- no oracle in between:
    shared and per-sample outer vectors, both row orders, the one-transform half, bit for bit."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_golden.npz"))
    x, s1, s2, gk = (g[f"{key}/{n}"] for n in ("x", "s1", "s2", "g"))
    S, B = gk.shape[0], x.shape[0] // gk.shape[0]
    bits = lambda v: v.view(np.uint8)   # noqa: E731
    for order, stride in (("batch", 1), ("sample", B)):
        kw = dict(axis="col", n_samples=S, sample_stride=stride)
        got = _hip.fused_shs(_t(x), _t(s1[0]), _t(gk), _t(s2[0]), **kw).cpu().numpy()
        assert np.array_equal(bits(got), bits(g[f"{key}/{order}/shared"])), (key, order)
        got = _hip.fused_shs(_t(x), _t(s1), _t(gk), _t(s2), a_per_sample=True, c_per_sample=True, **kw).cpu().numpy()
        assert np.array_equal(bits(got), bits(g[f"{key}/{order}/per_sample"])), (key, order)
        if _hip.fused_src_shared_supported(torch.from_numpy(x).dtype, x.shape[1]):
            got = _hip.fused_shs(_t(x), _t(s1[0]), _t(gk), None, one_transform=True, **kw).cpu().numpy()
            assert np.array_equal(bits(got), bits(g[f"{key}/{order}/one_transform"])), (key, order)


def test_integer_wrap_vectors_recorded_from_the_reference_gpu(hip_lib):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_golden.npz"))
    for D in (64, 4096):
        got = _hip.fwht_rows(_t(g[f"wrap_i32_D{D}/in"])).cpu().numpy()
        assert np.array_equal(got, g[f"wrap_i32_D{D}/out"]), D


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_partial_tiles_never_touch_memory_beyond_the_buffers(dtype, hip_lib):
    """The fused launch reaches a partial last tile (and every store) through bounds-checked buffer instructions whose
    chunk offset rides in the instruction's SCALAR offset -- an operand LLVM documents as excluded from the bounds check.
    On gfx950 the hardware check covers it; this test is what that claim rests on: source and destination are views in
    the middle of larger buffers filled with sentinels, row counts end mid-tile, and not one sentinel byte on either side
    of either buffer may change, in place and out of place."""
    torch.manual_seed(5)
    pad = 16384
    for d in (4, 16, 64, 256, 512, 2048, 4096):
        for rows in (1, 3, 5, 7, 33, 1000 // max(1, d // 64) + 1):
            n = rows * d
            big = torch.full((n + 2 * pad,), 12345.0, device=DEV, dtype=dtype)
            src_big = torch.full((n + 2 * pad,), 777.0, device=DEV, dtype=dtype)
            out, x = big[pad:pad + n].view(rows, d), src_big[pad:pad + n].view(rows, d)
            x.copy_(torch.randn(rows, d, device=DEV, dtype=dtype))
            a, c = (torch.randn(d, device=DEV, dtype=dtype) for _ in range(2))
            b = torch.randn(1, d, device=DEV, dtype=dtype)
            for in_place in (False, True):
                if in_place:
                    out.copy_(x)
                    _hip.fused_shs(out, a, b, c, axis="col", n_samples=1, sample_stride=rows, out=out)
                else:
                    _hip.fused_shs(x, a, b, c, axis="col", n_samples=1, sample_stride=rows, out=out)
                torch.cuda.synchronize()
                assert bool((big[:pad] == 12345.0).all()) and bool((big[pad + n:] == 12345.0).all()), (d, rows, in_place)
                assert bool((src_big[:pad] == 777.0).all()) and bool((src_big[pad + n:] == 777.0).all()), (d, rows, in_place)
                assert bool(torch.isfinite(out).all())
