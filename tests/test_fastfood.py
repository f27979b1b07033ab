"""Opt-in fastfood mode (whvi_amd/fastfood.py): the textbook operator S1 H diag(g) H S2 on activations -- the
Module-level consumer of the column-axis fused kernel (BASELINE config 3).  Not reference-equivalent by design
(SURVEY.md finding 1); its oracle is `oracle.pipeline(axis="col")`, a composition of the reference's own primitives,
and the dense product with the Hadamard matrix in float64 (the identity tests/test_oracle.py already pins)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
from whvi_amd.fastfood import FastfoodFunction, WHVIFastfoodMatrix
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

from test_host import ReplayRandn


def _layer(D, seed=0, bias=True):
    torch.manual_seed(seed)
    layer = WHVILinear(D, D, lambda_=0.7, bias=bias, mode="fastfood")
    with torch.no_grad():
        sub = layer.weight_submodule
        sub.g_mu.copy_(torch.randn(D) * 0.3)
        sub.s1.mul_(10.0)
        sub.s2.mul_(10.0)
        if bias:
            sub.bias.copy_(torch.randn(1, D) * 0.1)
    return layer


def _bits(a):
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def run_forward_parity(device, monkeypatch):
    D, S, B = 256, 5, 7
    layer = _layer(D).to(device)
    sub = layer.weight_submodule
    assert isinstance(sub, WHVIFastfoodMatrix)
    assert {n for n, _ in sub.named_parameters()} == {"bias", "s1", "s2", "g_mu", "g_rho"}
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, D)).astype(np.float32)
    eps = rng.standard_normal((S, D)).astype(np.float32)
    monkeypatch.setattr(torch, "randn", ReplayRandn([eps]))
    with torch.no_grad():
        y = layer.forward_mc(torch.from_numpy(x).to(device), S)
    monkeypatch.undo()
    assert y.shape == (S, B, D)
    s1, s2, bias = (t.detach().cpu().numpy() for t in (sub.s1, sub.s2, sub.bias))
    g = (sub.g_mu + sub.g_sigma * torch.from_numpy(eps).to(device)).detach().cpu().numpy()
    # rows in (sample, batch) order: sample_stride = B
    want = oracle.pipeline(np.tile(x, (S, 1)), s1, g, s2, n_samples=S, sample_stride=B, axis="col").reshape(S, B, D)
    got = y.cpu().numpy()
    assert np.array_equal(_bits(got), _bits((want + bias).astype(np.float32))), "bit-exact vs oracle.pipeline(col)"
    # the dense product in float64: y_k = x @ (diag(s1) H diag(g_k) H diag(s2)).T
    H = oracle.hadamard(D)
    for k in range(S):
        W = (s1.astype(np.float64)[:, None] * H) @ (g[k].astype(np.float64)[:, None] * (H * s2.astype(np.float64)[None, :]))
        ref = x.astype(np.float64) @ W.T + bias
        assert np.abs(got[k] - ref).max() <= 1e-5 * np.abs(ref).max()
        assert np.allclose(sub.dense_weight(torch.from_numpy(g[k]).to(device)).detach().cpu().numpy(), W, rtol=1e-4, atol=1e-4)
    # one-sample forward == sample 0 of the batched pass with the same eps
    monkeypatch.setattr(torch, "randn", ReplayRandn([eps[:1]]))
    with torch.no_grad():
        y1 = layer(torch.from_numpy(x).to(device))
    monkeypatch.undo()
    assert torch.equal(y1, y[0])
    # KL: the square layer's formula on the same parameters
    ref_layer = WHVILinear(D, D, lambda_=0.7, bias=True).to(device)
    ref_layer.load_state_dict(layer.state_dict())            # same names and shapes: checkpoints interchange
    assert torch.equal(ref_layer.kl, layer.kl)
    # ... and it is NOT the reference's (diagonal) operator
    monkeypatch.setattr(torch, "randn", ReplayRandn([eps[:1]]))
    with torch.no_grad():
        y_ref = ref_layer(torch.from_numpy(x).to(device))
    monkeypatch.undo()
    assert not torch.allclose(y_ref, y1, rtol=1e-2, atol=1e-3)


def run_gradients(device, dtype, monkeypatch, D=32):
    """Custom backward (fused launch with a and c exchanged + recomputed intermediates) against torch autograd over the
    dense float64 operator."""
    S, B = 3, 4
    g0 = torch.Generator().manual_seed(2)
    x = torch.randn(S * B, D, generator=g0, dtype=dtype).to(device).requires_grad_()
    a, c = (torch.randn(D, generator=g0, dtype=dtype).to(device).requires_grad_() for _ in range(2))
    b = torch.randn(S, D, generator=g0, dtype=dtype).to(device).requires_grad_()
    w = torch.randn(S * B, D, generator=g0, dtype=dtype).to(device)
    y = FastfoodFunction.apply(x, a, b, c, S, B)
    got = torch.autograd.grad((y * w).sum(), (x, a, b, c))
    H = torch.from_numpy(oracle.hadamard(D)).to(device)
    xd, ad, bd, cd = (t.detach().double().requires_grad_() for t in (x, a, b, c))
    rows = torch.arange(S * B, device=device) // B % S
    yd = ad * ((bd[rows] * ((cd * xd) @ H)) @ H)
    want = torch.autograd.grad((yd * w.double()).sum(), (xd, ad, bd, cd))
    assert float((y.double() - yd).abs().max()) <= (1e-5 if dtype == torch.float32 else 1e-12) * float(yd.abs().max())
    for name, p, q in zip("xabc", got, want):
        tol = (2e-5 if dtype == torch.float32 else 1e-12) * max(1.0, D / 32)
        assert float((p.double() - q).abs().max()) <= tol * float(q.abs().max()), name
    # x only (inference-time sensitivity): the single swapped launch
    y = FastfoodFunction.apply(x, a.detach(), b.detach(), c.detach(), S, B)
    gx, = torch.autograd.grad((y * w).sum(), (x,))
    assert float((gx.double() - want[0]).abs().max()) <= (2e-5 if dtype == torch.float32 else 1e-12) * max(1.0, D / 32) * float(want[0].abs().max())


def run_network(device):
    torch.manual_seed(0)
    net = WHVIRegression([nn.Linear(2, 16), nn.ReLU(), WHVILinear(16, 16, lambda_=1.0, mode="fastfood"), nn.ReLU(),
                          nn.Linear(16, 1)], train_samples=3, eval_samples=5).to(device)
    x, y = torch.randn(12, 2, device=device), torch.randn(12, 1, device=device)
    for mode in ("loop", "batched"):
        net.mc_mode = mode
        net.train()
        net.zero_grad()
        loss = net.loss(x, y, n=100)
        loss.backward()
        assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
        net.eval()
        assert net(x).shape == (12, 1, 5)


def test_forward_parity_cpu(monkeypatch):
    run_forward_parity("cpu", monkeypatch)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_gradients_cpu(dtype, monkeypatch):
    run_gradients("cpu", dtype, monkeypatch)


def test_network_cpu():
    run_network("cpu")


def test_mode_argument_validation():
    with pytest.raises(ValueError):
        WHVILinear(3, 16, mode="fastfood")
    with pytest.raises(ValueError):
        WHVILinear(8, 8, mode="textbook")
    assert type(WHVILinear(8, 8).weight_submodule).__name__ == "WHVISquarePow2Matrix"     # default = the reference


@pytest.mark.gpu
def test_forward_parity_gpu(monkeypatch, hip_lib):
    run_forward_parity("cuda", monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_gradients_gpu(dtype, monkeypatch, hip_lib):
    run_gradients("cuda", dtype, monkeypatch)


@pytest.mark.gpu
def test_network_gpu(hip_lib):
    run_network("cuda")


@pytest.mark.gpu
def test_config3_shape_through_the_module(monkeypatch, hip_lib):
    """BASELINE config 3's kernel shape through the Module: D = 2048, 64 MC samples, batch 512 (the production
    column-axis launch); 40 sampled rows bit-exact vs the oracle."""
    D, S, B = 2048, 64, 512
    layer = _layer(D, seed=3, bias=False).to("cuda")
    sub = layer.weight_submodule
    rng = np.random.default_rng(4)
    x = rng.standard_normal((B, D)).astype(np.float32)
    eps = rng.standard_normal((S, D)).astype(np.float32)
    monkeypatch.setattr(torch, "randn", ReplayRandn([eps]))
    with torch.no_grad():
        y = layer.forward_mc(torch.from_numpy(x).to("cuda"), S)
    monkeypatch.undo()
    g = (sub.g_mu + sub.g_sigma * torch.from_numpy(eps).to("cuda")).detach().cpu().numpy()
    idx = rng.integers(0, S * B, 40)
    got = y.reshape(S * B, D)[torch.from_numpy(idx).to("cuda")].cpu().numpy()
    want = oracle.pipeline(x[idx % B], sub.s1.detach().cpu().numpy(), g[idx // B], sub.s2.detach().cpu().numpy(),
                           n_samples=len(idx), sample_stride=1, axis="col")
    assert np.array_equal(_bits(got), _bits(want))


def run_small_widths(device, monkeypatch):
    """Rows shorter than one 16-byte chunk (D = 1, 2) up to one chunk / one quad / one row of lanes: the same module,
    bit-exact against the oracle."""
    for D in (1, 2, 4, 8, 16, 64):
        layer = _layer(D, seed=D, bias=False).to(device)
        sub = layer.weight_submodule
        rng = np.random.default_rng(D)
        S, B = 3, 5
        x = rng.standard_normal((B, D)).astype(np.float32)
        eps = rng.standard_normal((S, D)).astype(np.float32)
        monkeypatch.setattr(torch, "randn", ReplayRandn([eps]))
        with torch.no_grad():
            y = layer.forward_mc(torch.from_numpy(x).to(device), S).cpu().numpy()
        monkeypatch.undo()
        g = (sub.g_mu + sub.g_sigma * torch.from_numpy(eps).to(device)).detach().cpu().numpy()
        want = oracle.pipeline(np.tile(x, (S, 1)), sub.s1.detach().cpu().numpy(), g, sub.s2.detach().cpu().numpy(),
                               n_samples=S, sample_stride=B, axis="col").reshape(S, B, D)
        assert np.array_equal(_bits(y), _bits(want)), D


def test_small_widths_cpu(monkeypatch):
    run_small_widths("cpu", monkeypatch)


@pytest.mark.gpu
def test_small_widths_gpu(monkeypatch, hip_lib):
    run_small_widths("cuda", monkeypatch)


@pytest.mark.gpu
def test_layers_wider_than_the_fused_kernel_gpu(monkeypatch, hip_lib):
    """the constructor accepts any power of two but ``whvi_fused_shs`` stops at one wavefront tile
    (D = 8192); a D = 16384 layer used to build and then fail at its first forward.  It now runs the same multiplies and
    butterflies as separate launches (``fwht_rows``: a block per row) -- bit-exact against the same oracle, forward and
    backward both usable."""
    from whvi_amd import _hip
    D, S, B = 16384, 2, 3
    assert not _hip.fused_supported(torch.float32, D) and _hip.fused_supported(torch.float32, 8192)
    layer = _layer(D, seed=3, bias=False).to("cuda")
    sub = layer.weight_submodule
    rng = np.random.default_rng(6)
    x = rng.standard_normal((B, D)).astype(np.float32)
    eps = rng.standard_normal((S, D)).astype(np.float32)
    monkeypatch.setattr(torch, "randn", ReplayRandn([eps]))
    xt = torch.from_numpy(x).to("cuda").requires_grad_()
    y = layer.forward_mc(xt, S)
    monkeypatch.undo()
    g = (sub.g_mu + sub.g_sigma * torch.from_numpy(eps).to("cuda")).detach().cpu().numpy()
    want = oracle.pipeline(np.tile(x, (S, 1)), sub.s1.detach().cpu().numpy(), g, sub.s2.detach().cpu().numpy(),
                           n_samples=S, sample_stride=B, axis="col").reshape(S, B, D)
    assert np.array_equal(_bits(y.detach().cpu().numpy()), _bits(want))
    y.square().mean().backward()
    assert bool(torch.isfinite(xt.grad).all()) and all(bool(torch.isfinite(p.grad).all()) for p in sub.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_gradients_gpu_through_the_one_transform_launches(dtype, monkeypatch, hip_lib):
    """D = 512: rows long enough for the one-transform form of the fused kernel, which the backward pass then uses for
    each of its four "scale, then FWHT" steps; same check as above (autograd over the dense float64 operator) -- and a
    shared input's gradient is the sum over the samples."""
    from whvi_amd import _hip
    run_gradients("cuda", dtype, monkeypatch, D=512)
    assert "true>" in _hip.last_kernel() or "fused_shs_kernel" in _hip.last_kernel()
    D, S, B = 512, 3, 5
    g0 = torch.Generator().manual_seed(4)
    x = torch.randn(B, D, generator=g0, dtype=dtype).to("cuda").requires_grad_()
    a, c = (torch.randn(D, generator=g0, dtype=dtype).to("cuda").requires_grad_() for _ in range(2))
    b = torch.randn(S, D, generator=g0, dtype=dtype).to("cuda").requires_grad_()
    w = torch.randn(S * B, D, generator=g0, dtype=dtype).to("cuda")
    shared = torch.autograd.grad((FastfoodFunction.apply(x, a, b, c, S, B, True) * w).sum(), (x, a, b, c))
    x2 = x.detach().repeat(S, 1).requires_grad_()
    full = torch.autograd.grad((FastfoodFunction.apply(x2, a, b, c, S, B) * w).sum(), (x2, a, b, c))
    assert torch.allclose(shared[0], full[0].view(S, B, D).sum(dim=0), rtol=1e-5, atol=1e-6 * float(full[0].abs().max()))
    for p, q in zip(shared[1:], full[1:]):
        assert torch.equal(p, q)
