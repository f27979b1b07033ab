"""whvi_small_k_apply_f32 / whvi_row_dot_f32 (whvi_amd/csrc/layer_apply.hpp): the dense products of a stacked layer with a
narrow input (`x_padded @ W.T`, src/weights.py:179-180,195-206) and of a transposed column layer (`F.linear(x, w)`,
src/weights.py:239-251) for all Monte-Carlo samples per launch, against the torch.matmul forms they replace -- on arbitrary
operands, on the layers' as-written (diagonal) weights, with non-finite inputs, and through the Modules with gradients."""
import numpy as np
import pytest
import torch

from whvi_amd import _hip
from whvi_amd.layers import WHVILinear
from whvi_amd.weights import WHVIColumnMatrix, WHVIStackedMatrix

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("K,N,S,B", [(4, 1024, 16, 333), (4, 16, 3, 1000), (8, 64, 5, 77), (8, 1024, 2, 129), (4, 4, 2, 5),
                                     (4, 1024, 5, 16384 + 3), (4, 48, 3, 100), (8, 4096, 2, 50), (4, 3 * 1024, 2, 40)])
def test_small_k_apply_vs_matmul(K, N, S, B, hip_lib):
    g = torch.Generator(device=DEV).manual_seed(K * N + S)
    x = torch.randn(B, K, device=DEV, generator=g)
    w = torch.randn(S, N, K, device=DEV, generator=g)
    bias = torch.randn(1, N, device=DEV, generator=g)
    want = torch.matmul(x.double(), w.double().transpose(1, 2)) + bias.double()
    got = _hip.small_k_apply(x, w, bias)
    assert _hip.last_kernel().startswith("whvi::small_k_apply_kernel<float, ") and got.shape == (S, B, N)
    assert float((got.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    assert torch.equal(_hip.small_k_apply(x, w, bias, relu_out=True), torch.relu(got))
    assert float((_hip.small_k_apply(x, w).double() - (want - bias.double())).abs().max()) <= 2e-6 * float(want.abs().max())
    # block-diagonal weights (what the stacked layer builds: one non-zero per row of W): the single product, bit for bit,
    # and a non-finite input poisons its row through the exact zeros like the dense product does
    wd = torch.zeros_like(w)
    idx = torch.arange(N, device=DEV)
    wd[:, idx, idx % K] = w[:, idx, idx % K]
    x[1, 0], x[2, K - 1], x[3, 1] = float("inf"), float("nan"), float("-inf")
    x[4], x[0, 0] = 0.0, -0.0                                   # zero products of both signs: the GEMM's +0
    got, ref = _hip.small_k_apply(x, wd), torch.matmul(x, wd.transpose(1, 2))
    fin = torch.isfinite(ref)
    assert torch.equal(got[fin].view(torch.int32), ref[fin].view(torch.int32)), "bits, zeros included"
    na, nb = torch.isnan(got), torch.isnan(ref)
    assert bool((na == nb).all()) and bool((got[~na] == ref[~nb]).all())
    assert bool(na[:, 1, 1::K].all()) and bool(torch.isinf(got[:, 1, 0::K]).all()) and not bool(na[:, 0].any())      # x[1, 0] = inf


@pytest.mark.parametrize("D,S,B,relu", [(1024, 16, 257, False), (1024, 3, 90, True), (64, 5, 1000, False), (4, 2, 37, True),
                                        (4096, 2, 19, False), (512, 4, 40000 + 1, True)])
def test_row_dot_vs_matmul(D, S, B, relu, hip_lib):
    g = torch.Generator(device=DEV).manual_seed(D + S + B)
    x = torch.randn(S, B, D, device=DEV, generator=g)
    w = torch.randn(S, D, device=DEV, generator=g)
    bias = torch.randn(1, 1, device=DEV, generator=g)
    xin = torch.relu(x) if relu else x
    want = torch.matmul(xin.double(), w.double().unsqueeze(-1)) + bias.double()
    got = _hip.row_dot(x, w, bias, relu_in=relu)
    assert _hip.last_kernel().startswith("whvi::row_dot_kernel<float, ") and got.shape == (S, B, 1)
    scale = float((xin.double().abs() * w.double().abs().unsqueeze(1)).sum(dim=-1).max())       # sum of |terms|: the yardstick
    assert float((got.double() - want).abs().max()) <= 2e-6 * scale
    x[0, 1, D // 2], x[S - 1, B - 1, 0] = float("nan"), float("inf")
    got = _hip.row_dot(x, w)
    assert bool(torch.isnan(got[0, 1, 0])) and not bool(torch.isfinite(got[S - 1, B - 1, 0])) and bool(torch.isfinite(got[0, 0, 0]))


@pytest.mark.parametrize("n_in,n_out,bias", [(3, 64, False), (3, 1024, True), (5, 16, True), (4, 8, False)])
def test_stacked_layer_batched_pass_with_and_without_the_hip_product(n_in, n_out, bias, hip_lib, monkeypatch):
    """WHVIStackedMatrix.forward_mc: the HIP product == torch.matmul on the same weights -- value for value (the as-written
    sub-matrices are diagonal: one non-zero term per sum) -- and so are the gradients to 1e-6."""
    torch.manual_seed(n_in * 100 + n_out)
    layer = WHVILinear(n_in, n_out, bias=bias).to(DEV)
    assert isinstance(layer.weight_submodule, WHVIStackedMatrix)
    with torch.no_grad():
        for name, p in layer.named_parameters():
            p.mul_(20.0) if name.endswith(("s1", "s2")) else p.normal_() if name.endswith(("g_mu", "bias")) else None
    x = torch.randn(50, n_in, device=DEV, requires_grad=True)
    outs = {}
    for flag in (True, False):
        monkeypatch.setattr(WHVIStackedMatrix, "hip_apply", flag)
        torch.manual_seed(9)
        layer.zero_grad(set_to_none=True)
        x.grad = None
        y = layer.forward_mc(x, 6)
        assert ("small_k_apply" in _hip.last_kernel()) == flag
        (y.square().sum() + layer.kl).backward()
        outs[flag] = (y.detach(), x.grad.clone(), [p.grad.clone() for p in layer.parameters()])
    assert outs[True][0].shape == (6, 50, n_out) and torch.equal(outs[True][0], outs[False][0])
    for a, b in zip([outs[True][1]] + outs[True][2], [outs[False][1]] + outs[False][2]):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-30


@pytest.mark.parametrize("n_in,bias", [(1024, False), (64, True), (16, True)])
def test_transposed_column_layer_batched_pass_with_and_without_the_hip_dot(n_in, bias, hip_lib, monkeypatch):
    torch.manual_seed(n_in)
    layer = WHVILinear(n_in, 1, bias=bias).to(DEV)
    assert isinstance(layer.weight_submodule, WHVIColumnMatrix) and layer.weight_submodule.transposed
    with torch.no_grad():
        for name, p in layer.named_parameters():
            p.mul_(20.0) if name.endswith(("s1", "s2")) else p.normal_() if name.endswith(("g_mu", "bias")) else None
    x = torch.randn(4, 33, n_in, device=DEV, requires_grad=True)
    outs = {}
    for flag in (True, False):
        monkeypatch.setattr(WHVIColumnMatrix, "hip_apply", flag)
        torch.manual_seed(3)
        layer.zero_grad(set_to_none=True)
        x.grad = None
        y = layer.forward_mc(x, 4)
        assert ("row_dot" in _hip.last_kernel()) == flag
        (y.square().sum() + layer.kl).backward()
        outs[flag] = (y.detach(), x.grad.clone(), [p.grad.clone() for p in layer.parameters()])
    assert outs[True][0].shape == (4, 33, 1)
    for a, b in zip([outs[True][0], outs[True][1]] + outs[True][2], [outs[False][0], outs[False][1]] + outs[False][2]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-30


def test_argument_checks(hip_lib):
    x = torch.randn(8, 4, device=DEV)
    w = torch.randn(2, 16, 4, device=DEV)
    with pytest.raises(RuntimeError, match="unsupported"):
        _hip.small_k_apply(torch.randn(8, 3, device=DEV), torch.randn(2, 16, 3, device=DEV))
    with pytest.raises(RuntimeError, match="unsupported"):
        _hip.row_dot(torch.randn(2, 8, 48, device=DEV), torch.randn(2, 48, device=DEV))
    fn = _hip.lib().whvi_small_k_apply_f32
    out = torch.empty(2, 8, 16, device=DEV)
    assert fn(out.data_ptr(), x.data_ptr(), w.data_ptr(), None, 2, 8, 16, 4, 0, None) == -2          # K = 16
    assert fn(out.data_ptr(), x.data_ptr(), w.data_ptr(), None, 2, 8, 18, 2, 0, None) == -1          # N % 4
    assert fn(out.data_ptr(), x.data_ptr(), w.data_ptr(), None, 2, 8, 4 * 7 * 5, 2, 0, None) == -2   # 35 column groups: 5 per thread
    assert fn(out.data_ptr(), x.data_ptr(), w.data_ptr(), None, 2, 8, 16, 2, 7, None) == -1          # unknown flags
    assert fn(None, None, None, None, 0, 8, 16, 2, 0, None) == 0
    rd = _hip.lib().whvi_row_dot_f32
    assert rd(out.data_ptr(), x.data_ptr(), w.data_ptr(), None, 2, 8, 13, 0, None) == -2
    assert rd(out.data_ptr(), x.data_ptr(), w.data_ptr(), None, 2, 8, 4, 2, None) == -1


def test_relu_folded_into_the_stacked_and_column_launches(monkeypatch, hip_lib):
    """A 3 -> N -> 1 network (stacked layer, nn.ReLU, transposed column layer -- no square layer in between): the activation is
    folded into the stacked layer's product (applied before the store).  Same loss and gradients as with the activation as a
    pass of its own; and each layer's ``forward_mc(relu_in=, relu_out=)`` equals torch.relu around the plain call."""
    import torch.nn as nn
    import whvi_amd.networks as networks
    from whvi_amd.networks import WHVIRegression
    torch.manual_seed(5)
    net = WHVIRegression([WHVILinear(3, 64, bias=True), nn.ReLU(), WHVILinear(64, 1, bias=True)], train_samples=4, eval_samples=4)
    with torch.no_grad():
        for name, p in net.named_parameters():
            p.mul_(30.0) if name.endswith(("s1", "s2")) else p.normal_() if name.endswith(("g_mu", "bias")) else None
    net = net.to(DEV).train()
    x, y = torch.randn(50, 3, device=DEV), torch.randn(50, 1, device=DEV)
    calls = []
    real = WHVIStackedMatrix.forward_mc
    monkeypatch.setattr(WHVIStackedMatrix, "forward_mc",
                        lambda self, x_, n, relu_in=False, relu_out=False: (calls.append((relu_in, relu_out)), real(self, x_, n, relu_in, relu_out))[1])

    def run():
        torch.manual_seed(2)
        net.zero_grad(set_to_none=True)
        loss = net.loss(x, y, n=500)
        loss.backward()
        return loss.detach(), [p.grad.clone() for p in net.parameters()]
    fused = run()
    assert calls == [(False, True)], calls
    monkeypatch.setattr(networks, "_fuses_relu", lambda module, h: False)
    plain = run()
    assert calls[-1] == (False, False)
    assert abs(float(fused[0]) - float(plain[0])) <= 1e-6 * abs(float(plain[0]))
    for a, b in zip(fused[1], plain[1]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-30
    monkeypatch.undo()
    stacked, column = net.sequential[0], net.sequential[2]
    h = torch.randn(4, 50, 64, device=DEV)
    with torch.no_grad():
        for layer, inp, kw in ((stacked, x, dict(relu_out=True)), (stacked, x, dict(relu_in=True, relu_out=True)),
                               (column, h, dict(relu_in=True)), (column, h, dict(relu_in=True, relu_out=True))):
            torch.manual_seed(1)
            got = layer.forward_mc(inp, 4, **kw)
            torch.manual_seed(1)
            want = layer.forward_mc(torch.relu(inp) if kw.get("relu_in") else inp, 4)
            want = torch.relu(want) if kw.get("relu_out") else want
            assert torch.equal(got, want), kw
