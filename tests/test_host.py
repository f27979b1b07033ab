"""Host-side mirror of the reference interface (whvi_amd/*) on the CPU, against golden bundles
recorded from the live reference and against the reference's own unit-test assertions."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import fwht_cpp
from whvi_amd.fwht import cpp as cpp_fwht
from whvi_amd.fwht import python as python_fwht
from whvi_amd.layers import WHVILinear
from whvi_amd.likelihoods import GaussianLikelihood
from whvi_amd.networks import WHVIRegression
from whvi_amd.utils import build_H, kl_diag_normal, matmul_diag_left, matmul_diag_right
from whvi_amd.weights import WHVIStackedMatrix, WHVISquarePow2Matrix

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def fg():
    return np.load(os.path.join(GOLD, "fwht_golden.npz"))


class ReplayRandn:
    """Replays recorded ``torch.randn`` draws in order (the reference draws eps inside forward,
    src/weights.py:82,92); anything else falls through to the real generator."""

    def __init__(self, draws, device=None):
        self.draws, self.i, self.real, self.device = list(draws), 0, torch.randn, device

    def __call__(self, *a, **k):
        if self.i < len(self.draws):
            size = a[0] if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else a
            want = int(np.prod(size)) if len(size) else 1
            first = np.array(self.draws[self.i])
            m = max(1, want // max(1, first.size))
            # a batched draw of shape (m, D) stands for m of the reference's sequential randn(D)
            block = np.stack([np.array(d) for d in self.draws[self.i:self.i + m]]) if m > 1 else first
            self.i += m
            t = torch.from_numpy(np.ascontiguousarray(block)).reshape(tuple(size))
            dev = k.get("device", None)
            return t.to(dev) if dev is not None else t
        return self.real(*a, **k)


# ---- FWHT front-ends -------------------------------------------------------------------------
@pytest.mark.parametrize("d", [1, 2, 4, 8, 32, 64, 512, 1024, 4096])
def test_cpp_and_python_front_ends_bit_equal_to_reference(fg, d):
    x = torch.from_numpy(fg[f"f32_in_{d}"])
    want = torch.from_numpy(fg[f"f32_out_{d}"])
    assert torch.equal(cpp_fwht.FWHTFunction.apply(x), want)
    assert torch.equal(python_fwht.FWHTFunction.apply(x), want)
    assert torch.equal(cpp_fwht.FWHT()(x), want)
    xi = torch.from_numpy(fg[f"i32_in_{d}"])
    assert torch.equal(fwht_cpp.forward(xi), torch.from_numpy(fg[f"i32_out_{d}"]))
    assert torch.equal(fwht_cpp.forward(xi.long()), torch.from_numpy(fg[f"i32_out_{d}"]).long())
    xd = torch.from_numpy(fg[f"f64_in_{d}"])
    assert torch.equal(fwht_cpp.backward(xd), torch.from_numpy(fg[f"f64_out_{d}"]))
    assert torch.equal(x, torch.from_numpy(fg[f"f32_in_{d}"])), "input must not be modified"


def test_config1_shape_on_the_host():
    """BASELINE config 1 (benchmarks/walsh.py shape on the CPU: D = 512, batch 1024, fp32): the host library through the
    reference's front-end classes, bit-equal to the oracle and -- where it is present -- to the reference's own compiled
    C++ FWHT (oracle/_ref); integer data additionally satisfies H.H = 512.I exactly."""
    import oracle
    g = torch.Generator().manual_seed(512)
    x = torch.randn(1024, 512, generator=g)
    want = torch.from_numpy(oracle.fwht(x.numpy()))
    for got in (cpp_fwht.FWHTFunction.apply(x), python_fwht.FWHTFunction.apply(x), fwht_cpp.forward(x)):
        assert torch.equal(got, want)
    ref = oracle.load_reference_cpp()
    if ref is not None:
        assert torch.equal(ref.forward(x), want)
    xi = torch.randint(-8, 8, (1024, 512), generator=g, dtype=torch.int32)
    assert torch.equal(fwht_cpp.forward(fwht_cpp.forward(xi)), xi * 512)


def test_cpp_front_end_3d_input_like_benchmarks(fg):
    x = torch.from_numpy(fg["f32_in_3d"])     # benchmarks/walsh.py:21 feeds (1, D, D)
    assert torch.equal(fwht_cpp.forward(x), torch.from_numpy(fg["f32_out_3d"]))


def test_reference_walsh_suite_cpu_cases():
    """test/walsh.py:11-59 re-stated against this package's front-ends."""
    a = torch.tensor([[1.0], [2.0], [3.0], [4.0]]).T
    assert torch.allclose(cpp_fwht.FWHTFunction.apply(a), torch.tensor([[10.0], [-2.0], [-4.0], [0.0]]).T, atol=1e-5)
    a = torch.tensor([[0.0], [1.0], [2.0], [3.0]]).T
    assert torch.allclose(cpp_fwht.FWHTFunction.apply(a), torch.tensor([[6.0], [-2.0], [-4.0], [0.0]]).T, atol=1e-5)
    D = 2 ** 5
    H = build_H(D, torch.device("cpu"))
    g = torch.Generator().manual_seed(0)
    for batch in (1, 40):
        for _ in range(30):
            A = torch.randn(batch, D, generator=g)
            reference = (H @ A.T).T
            assert torch.allclose(cpp_fwht.FWHTFunction.apply(A), reference, atol=1e-5)
            assert torch.allclose(python_fwht.WHT_matmul().apply(A), reference)
            assert torch.allclose(cpp_fwht.FWHTFunction.apply(A), python_fwht.FWHTFunction.apply(A), atol=1e-3)


def test_gradcheck_host_front_ends():
    # src/fwht/grad_check.py:26-34
    for fn in (cpp_fwht.FWHTFunction.apply, python_fwht.FWHTFunction.apply):
        x = torch.randn(3, 32, dtype=torch.float64, requires_grad=True)
        assert torch.autograd.gradcheck(fn, (x,))


# ---- utils -----------------------------------------------------------------------------------
def test_utils_against_reference_values(fg):
    A, d = torch.from_numpy(fg["diag_A"]), torch.from_numpy(fg["diag_d"])
    assert torch.equal(matmul_diag_left(d, A), torch.from_numpy(fg["diag_left"]))
    assert torch.equal(matmul_diag_right(A, d), torch.from_numpy(fg["diag_right"]))
    assert torch.allclose(torch.diag(d) @ A, matmul_diag_left(d, A))           # test/utils.py:8-13
    assert torch.allclose(A @ torch.diag(d), matmul_diag_right(A, d))           # test/utils.py:15-20
    assert torch.equal(build_H(8, torch.device("cpu")), torch.from_numpy(fg["H_8"]))
    mu1, sd1, mu2, sd2 = (torch.from_numpy(v) for v in fg["kl_args"])
    assert torch.allclose(kl_diag_normal(mu1, sd1, mu2, sd2), torch.from_numpy(fg["kl_value"]), rtol=1e-6)
    want = torch.distributions.kl.kl_divergence(                                 # test/utils.py:22-34
        torch.distributions.MultivariateNormal(mu1, torch.diag(sd1)),
        torch.distributions.MultivariateNormal(mu2, torch.diag(sd2)))
    assert torch.allclose(want, kl_diag_normal(mu1, sd1, mu2, sd2))


# ---- layers ----------------------------------------------------------------------------------
def _bundle(name):
    g = np.load(os.path.join(GOLD, "whvi_golden.npz"))
    return {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(name + "/")}


def _layer_from_bundle(b, device="cpu"):
    layer = WHVILinear(int(b["n_in"]), int(b["n_out"]), lambda_=float(b["lambda_"]), bias=bool(int(b["bias"])))
    state = {k[len("param."):]: torch.from_numpy(np.array(v)) for k, v in b.items() if k.startswith("param.")}
    assert set(state) == set(dict(layer.named_parameters())), "parameter names must match the reference"
    layer.load_state_dict(state)
    return layer.to(device)


def run_layer_bundle(name, device, monkeypatch, rtol):
    b = _bundle(name)
    layer = _layer_from_bundle(b, device)
    eps = [b[f"eps{i}"] for i in range(int(b["n_eps"]))]
    x = torch.from_numpy(b["x"]).to(device).requires_grad_(True)
    weight = torch.from_numpy(b["weight"]).to(device)
    replay = ReplayRandn(eps)
    monkeypatch.setattr(torch, "randn", replay)
    y = layer(x)
    monkeypatch.undo()
    assert replay.i == len(eps), "must draw exactly the reference's number of eps vectors"
    kl = layer.kl
    ((y * weight).sum() + kl).backward()

    def close(got, want, what):
        want = torch.from_numpy(np.array(want))
        scale = float(want.abs().max()) or 1.0
        err = float((got.detach().cpu() - want).abs().max())
        assert err <= rtol * scale, f"{name}: {what} err {err:.3e} > {rtol} * {scale:.3e}"

    close(y, b["y"], "forward")
    close(kl, b["kl"], "kl")
    close(x.grad, b["grad_x"], "grad_x")
    for pname, p in layer.named_parameters():
        close(p.grad, b["grad." + pname], "grad " + pname)


@pytest.mark.parametrize("name", ["sq8", "sq64b", "sq512", "sq4096", "st3x16", "st5x7b", "st13x128",
                                  "col1x10b", "col16x1"])
def test_layer_forward_kl_backward_vs_reference_cpu(name, monkeypatch):
    # same torch ops as the reference on the host -> far inside the 1e-5 bar
    run_layer_bundle(name, "cpu", monkeypatch, rtol=1e-5)


def test_dispatch_and_parameter_layout():
    from whvi_amd.weights import WHVIColumnMatrix
    assert isinstance(WHVILinear(1, 10).weight_submodule, WHVIColumnMatrix)
    assert WHVILinear(16, 1).weight_submodule.transposed
    assert isinstance(WHVILinear(8, 8).weight_submodule, WHVISquarePow2Matrix)
    st = WHVILinear(3, 1024).weight_submodule
    assert isinstance(st, WHVIStackedMatrix) and (st.D_in, st.D_out, st.padding, st.stack) == (4, 1024, 1, 256)
    sq = WHVISquarePow2Matrix(64, bias=True)
    assert {n: tuple(p.shape) for n, p in sq.named_parameters()} == {
        "bias": (1, 64), "s1": (64,), "s2": (64,), "g_mu": (64,), "g_rho": (64,)}
    assert float(sq.g_mu.abs().max()) == 0.0 and -3.0 <= float(sq.g_rho.min()) and float(sq.g_rho.max()) <= -2.0
    assert float(sq.s1.abs().max()) < 0.06     # 0.01 * N(0, 1)


def test_same_seed_same_initialisation_as_reference():
    """Parameter creation consumes the RNG in the reference's order, so a seed reproduces the
    reference's initial state_dict (values recorded in the golden bundle before perturbation are
    not stored; here: self-consistency of the order bias, s1, s2, g_mu, g_rho)."""
    torch.manual_seed(3)
    a = WHVISquarePow2Matrix(16, bias=True)
    torch.manual_seed(3)
    s1 = torch.randn(16) * 0.01
    s2 = torch.randn(16) * 0.01
    g_rho = torch.rand(16) - 3
    assert torch.equal(a.s1.data, s1) and torch.equal(a.s2.data, s2) and torch.equal(a.g_rho.data, g_rho)


# ---- network + likelihood --------------------------------------------------------------------
def test_network_shapes_like_reference_test():
    # test/networks.py:11-23
    for k in range(1, 21):
        net = WHVIRegression([nn.Linear(1, 8), nn.ReLU(), WHVILinear(8, 8), nn.ReLU(), nn.Linear(8, k)],
                             train_samples=5, eval_samples=6)
        net.train()
        assert net(torch.randn(50, 1)).size() == (50, k, 5)
        net.eval()
        assert net(torch.randn(50, 1)).size() == (50, k, 6)


def test_network_values_vs_reference(monkeypatch):
    g = np.load(os.path.join(GOLD, "network_golden.npz"))
    net = WHVIRegression([nn.Linear(1, 8), nn.ReLU(), WHVILinear(8, 8), nn.ReLU(), nn.Linear(8, 2)],
                         train_samples=3, eval_samples=4)
    state = {k[len("state."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state.")}
    assert set(state) == set(net.state_dict()), "checkpoint keys must interchange with the reference"
    net.load_state_dict(state)
    net.train()
    monkeypatch.setattr(torch, "randn", ReplayRandn([g[f"eps{i}"] for i in range(int(g["n_eps"]))]))
    pred = net(torch.from_numpy(g["x"]))
    monkeypatch.undo()
    assert torch.allclose(pred, torch.from_numpy(g["pred"]), rtol=1e-5, atol=1e-6)
    mnll = net.likelihood.mnll_batch_estimate(torch.from_numpy(g["y"]), pred, 100)
    assert torch.allclose(mnll, torch.from_numpy(g["mnll"]), rtol=1e-5)
    assert torch.allclose(net.kl, torch.from_numpy(g["kl"]), rtol=1e-6)


def test_gaussian_likelihood_like_reference_tests():
    # test/likelihoods.py:8-56 (explicit double loop, delta 1e-4)
    g = np.load(os.path.join(GOLD, "network_golden.npz"))
    y = torch.reshape(torch.tensor([0., 1., 2., -1.]), (-1, 1))
    y_hat = torch.tensor([[0.2, 1.1, 2.2, -1.3], [-0.1, 1.05, 2, -1.1]]).T.unsqueeze(1)
    got = float(GaussianLikelihood(sigma=1.0).mnll_batch_estimate(y, y_hat, 12))
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
    assert abs(target - float(GaussianLikelihood(sigma=sigma).mnll_batch_estimate(y, y_hat, n))) < 1e-4


def test_train_and_eval_loop_runs(tmp_path):
    from torch.utils.data import DataLoader, TensorDataset
    torch.manual_seed(0)
    X, Y = torch.randn(32, 3), torch.randn(32, 1)
    net = WHVIRegression([WHVILinear(3, 8), nn.ReLU(), WHVILinear(8, 8), nn.ReLU(), WHVILinear(8, 1)], eval_samples=4)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda t: 1.0)
    net.train_model(DataLoader(TensorDataset(X, Y), batch_size=16), opt, sched, epochs1=2, epochs2=2,
                    checkpoint_dir=tmp_path)
    assert (tmp_path / "epoch-0.pth").exists()
    err, mnll = net.eval_model(X, Y)
    assert np.isfinite(err) and np.isfinite(mnll)


# ---- batched Monte-Carlo pass (F1) vs the reference-style loop ---------------------------------
class EpsRouter:
    """Serves ``torch.randn`` from per-layer tables ``E[D]`` of shape (S, J, D) so that the loop
    (one draw per sample per sub-matrix) and the batched pass (one draw per layer) see the SAME eps
    for every (layer, sample, sub-matrix).  Layers are told apart by their D."""

    def __init__(self, tables, mode, real):
        self.tables, self.mode, self.real, self.count = tables, mode, real, {}

    def __call__(self, *a, **k):
        size = tuple(a[0]) if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else tuple(a)
        D = size[-1]
        if D not in self.tables:
            return self.real(*a, **k)
        E = self.tables[D]
        n = self.count.get(D, 0)
        self.count[D] = n + 1
        if len(size) == 1:                                   # loop, one sub-matrix of one sample
            s, j = divmod(n, E.shape[1])
            out = E[s, j]
        elif len(size) == 3:                                 # batched stacked: (J, S, D)
            out = E.transpose(0, 1)
        elif self.mode == "loop":                            # loop on the GPU, stacked: (J, D) of sample n
            out = E[n]
        else:                                                # batched square / column: (S, D)
            out = E[:, 0]
        assert tuple(out.shape) == size, (out.shape, size)
        dev = k.get("device", None)
        return out.clone().to(dev) if dev is not None else out.clone()


def loop_vs_batched(device, monkeypatch):
    torch.manual_seed(4)
    S = 5
    net = WHVIRegression([WHVILinear(3, 16, bias=True), nn.ReLU(), WHVILinear(16, 16), nn.ReLU(), nn.Linear(16, 8),
                          nn.ReLU(), WHVILinear(8, 1, bias=True)], train_samples=S).to(device)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith("g_mu") or n_.endswith("s1") or n_.endswith("s2"):
                p_.copy_(torch.randn(p_.shape) * 0.5)
    g = torch.Generator().manual_seed(8)
    tables = {4: torch.randn(S, 4, 4, generator=g), 16: torch.randn(S, 1, 16, generator=g),
              8: torch.randn(S, 1, 8, generator=g)}
    x, y = torch.randn(9, 3, generator=g).to(device), torch.randn(9, 1, generator=g).to(device)
    outs, grads = {}, {}
    real = torch.randn
    for mode in ("loop", "batched"):
        net.mc_mode = mode
        net.zero_grad()
        monkeypatch.setattr(torch, "randn", EpsRouter(tables, mode, real))
        loss = net.loss(x, y, n=90)          # batched mode on the GPU takes the KL from the fused pass
        monkeypatch.undo()
        loss.backward()
        losses = locals().setdefault("losses", {})
        losses[mode] = float(loss)
        net.zero_grad()
        monkeypatch.setattr(torch, "randn", EpsRouter(tables, mode, real))
        pred = net(x)
        monkeypatch.undo()
        (net.likelihood.mnll_batch_estimate(y, pred, 90) + net.kl).backward()
        outs[mode] = pred.detach().cpu()
        grads[mode] = torch.cat([p_.grad.reshape(-1) for p_ in net.parameters()]).cpu()
    assert outs["loop"].shape == (9, 1, S)
    scale = float(outs["loop"].abs().max())
    assert float((outs["loop"] - outs["batched"]).abs().max()) <= 1e-5 * scale
    gscale = float(grads["loop"].abs().max())
    assert float((grads["loop"] - grads["batched"]).abs().max()) <= 1e-4 * gscale
    assert abs(losses["loop"] - losses["batched"]) <= 1e-5 * abs(losses["loop"])


def test_batched_mc_pass_equals_loop_cpu(monkeypatch):
    loop_vs_batched("cpu", monkeypatch)


def test_batched_mc_pass_reference_bundle(monkeypatch):
    """The recorded reference run (3 samples through one WHVILinear(8,8)) through the batched pass."""
    g = np.load(os.path.join(GOLD, "network_golden.npz"))
    net = WHVIRegression([nn.Linear(1, 8), nn.ReLU(), WHVILinear(8, 8), nn.ReLU(), nn.Linear(8, 2)],
                         train_samples=3, eval_samples=4)
    net.load_state_dict({k[len("state."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state.")})
    net.train()
    net.mc_mode = "batched"
    monkeypatch.setattr(torch, "randn", ReplayRandn([g[f"eps{i}"] for i in range(int(g["n_eps"]))]))
    pred = net(torch.from_numpy(g["x"]))
    monkeypatch.undo()
    assert torch.allclose(pred, torch.from_numpy(g["pred"]), rtol=1e-5, atol=1e-6)


def test_src_alias_package_matches_reference_import_paths():
    """``src.*`` (the reference's import paths, README.md:30-33 of the reference) resolve to this package."""
    import importlib
    import whvi_amd.layers
    import whvi_amd.networks
    assert importlib.import_module("src.layers").WHVILinear is whvi_amd.layers.WHVILinear
    assert importlib.import_module("src.networks").WHVIRegression is whvi_amd.networks.WHVIRegression
    for mod, names in (("src.utils", ["matmul_diag_left", "matmul_diag_right", "kl_diag_normal", "build_H", "is_pow_of_2"]),
                       ("src.weights", ["WHVISquarePow2Matrix", "WHVIStackedMatrix", "WHVIColumnMatrix"]),
                       ("src.likelihoods", ["GaussianLikelihood", "Likelihood"]),
                       ("src.activations", ["Cosine"]),
                       ("src.evaluation", ["make_optimizer", "evaluate_bayesian_regression_dnn"]),
                       ("src.fwht.cpp.fwht", ["FWHTFunction", "FWHT"]), ("src.fwht.cuda.fwht", ["FWHTFunction"]),
                       ("src.fwht.python.fwht", ["FWHTFunction", "WHT_matmul", "FWHT"])):
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)


# ---- opt-in packed parameter layout of stacked layers ---------------------------------------------------
@pytest.mark.parametrize("n_in,n_out", [(3, 16), (5, 7), (13, 128), (100, 33)])
def test_packed_stacked_layer_is_the_same_layer(n_in, n_out, monkeypatch):
    """``pack_parameters()``: 4 parameter tensors instead of 4 * stack, identical forward / KL / gradients for the
    same eps, and ``state_dict`` in the reference's per-sub-matrix format both ways (also after deepcopy / pickle)."""
    import copy
    import io
    torch.manual_seed(n_in + n_out)
    plain = WHVILinear(n_in, n_out, lambda_=0.5, bias=True)
    with torch.no_grad():
        for name, p in plain.named_parameters():
            if name.endswith("g_mu") or name.endswith("s1") or name.endswith("s2"):
                p.copy_(torch.randn(p.shape) * 0.4)
    packed = copy.deepcopy(plain)
    packed.weight_submodule.pack_parameters()
    sub = packed.weight_submodule
    assert len(list(packed.parameters())) == 5 and len(list(plain.parameters())) == 4 * sub.stack + 1
    assert list(packed.state_dict().keys()) and set(packed.state_dict()) == set(plain.state_dict())
    for k, v in plain.state_dict().items():
        assert torch.equal(packed.state_dict()[k], v), k
    eps = [np.random.default_rng(3 + i).standard_normal(sub.D_in).astype(np.float32) for i in range(sub.stack)]
    x = torch.randn(6, n_in)
    outs = []
    for layer in (plain, packed):
        for use_mc in (False, True):
            layer.zero_grad()
            monkeypatch.setattr(torch, "randn", ReplayRandn(eps))
            y = layer.weight_submodule.forward_mc(x, 1)[0] if use_mc else layer(x)
            monkeypatch.undo()
            (y.square().sum() + layer.kl).backward()
            sd = layer.state_dict(keep_vars=True)
            grads = {k: None for k in sd}
            if layer is plain:
                grads = {k: v.grad.clone() for k, v in sd.items()}
            else:
                for name in ("s1", "s2", "g_mu", "g_rho"):
                    g = getattr(sub, "packed_" + name).grad
                    for j in range(sub.stack):
                        grads[f"weight_submodule.weight_matrices.{j}.{name}"] = g[j].clone()
                grads["weight_submodule.bias"] = sub.bias.grad.clone()
            outs.append((y.detach().clone(), float(layer.kl), grads))
    for a, b in ((outs[0], outs[2]), (outs[1], outs[3])):
        assert torch.equal(a[0], b[0]) and a[1] == b[1]
        for k in a[2]:
            assert torch.allclose(a[2][k], b[2][k], rtol=1e-6, atol=1e-7), k
    # checkpoints interchange in both directions, strictly
    fresh_plain = WHVILinear(n_in, n_out, lambda_=0.5, bias=True)
    fresh_plain.load_state_dict(packed.state_dict())
    fresh_packed = WHVILinear(n_in, n_out, lambda_=0.5, bias=True)
    fresh_packed.weight_submodule.pack_parameters()
    fresh_packed.load_state_dict(plain.state_dict())
    for k, v in plain.state_dict().items():
        assert torch.equal(fresh_plain.state_dict()[k], v) and torch.equal(fresh_packed.state_dict()[k], v)
    with pytest.raises(RuntimeError, match="Missing key"):
        fresh_packed.load_state_dict({k: v for k, v in plain.state_dict().items() if not k.endswith("0.s1")})
    # deepcopy and pickle keep the sub-matrix views bound to their own parent
    twin = copy.deepcopy(packed)
    with torch.no_grad():
        twin.weight_submodule.packed_s1.add_(1.0)
    assert torch.equal(twin.weight_submodule.weight_matrices[0].s1, twin.weight_submodule.packed_s1[0])
    assert not torch.equal(twin.weight_submodule.weight_matrices[0].s1, packed.weight_submodule.weight_matrices[0].s1)
    buf = io.BytesIO()
    torch.save(packed, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    last = sub.stack - 1
    assert torch.equal(loaded.weight_submodule.weight_matrices[last].g_rho, packed.weight_submodule.packed_g_rho[last])
    assert loaded(x * 0).shape == (6, n_out) and torch.equal(loaded.weight_submodule.packed_s2, packed.weight_submodule.packed_s2)


def test_fast_training_path_refuses_the_host_clearly(tmp_path):
    """``make_optimizer(capturable=True)`` and ``train_model(graphed=True)`` are GPU features: on host tensors they say so
    instead of failing somewhere inside torch; ``packed=True`` after the optimizer exists is refused with the remedy; the
    plain recipe is untouched by the new keywords (same trajectory with and without them spelled out)."""
    import copy
    import torch.nn as nn
    from torch.utils.data import DataLoader, TensorDataset
    from whvi_amd.evaluation import make_optimizer
    from whvi_amd.layers import WHVILinear
    from whvi_amd.networks import WHVIRegression
    torch.manual_seed(0)
    net = WHVIRegression([WHVILinear(3, 8), nn.ReLU(), WHVILinear(8, 1)], train_samples=2)
    twin = copy.deepcopy(net)
    x, y = torch.randn(12, 3), torch.randn(12, 1)
    loader = DataLoader(TensorDataset(x, y), batch_size=6)
    with pytest.raises(RuntimeError, match="needs the network on a GPU"):
        make_optimizer(net, capturable=True)
    optimizer, scheduler = make_optimizer(net, lambda0=0.05)
    with pytest.raises(RuntimeError, match="needs GPU tensors"):
        net.train_model(loader, optimizer, scheduler, epochs1=1, epochs2=0, graphed=True)
    with pytest.raises(RuntimeError, match="BEFORE creating the optimizer"):
        net.train_model(loader, optimizer, scheduler, epochs1=1, epochs2=0, packed=True)
    # the reference's call and the same call with every new keyword at its default walk the same trajectory
    opt2, sched2 = make_optimizer(twin, lambda0=0.05, capturable=False, packed=False)
    torch.manual_seed(5)
    net.train_model(loader, optimizer, scheduler, epochs1=1, epochs2=1, checkpoint_dir=tmp_path)
    torch.manual_seed(5)
    twin.train_model(loader, opt2, sched2, epochs1=1, epochs2=1, graphed=False, packed=None, sharded=False)
    for (k, a), (_, b) in zip(net.named_parameters(), twin.named_parameters()):
        assert torch.equal(a, b), k
    # packed layout chosen through make_optimizer: 4 tensors per stacked layer, the reference's checkpoint keys
    third = WHVIRegression([WHVILinear(3, 8), nn.ReLU(), WHVILinear(8, 1)], train_samples=2)
    keys = list(third.state_dict().keys())
    opt3, sched3 = make_optimizer(third, packed=True)
    assert len(list(third.parameters())) < len(keys) and list(third.state_dict().keys()) == keys
    third.train_model(loader, opt3, sched3, epochs1=1, epochs2=0, packed=True)


def test_evaluation_harness_protocol_on_the_host(tmp_path, capsys):
    """``evaluate_bayesian_regression_dnn`` (src/evaluation.py:30-108) with the reference's flow (``fast=False``: DataLoader,
    host Adam + LambdaLR): standardised inputs, 90 / 10 splits, the (n_in, hidden, hidden, n_out) network with lambda_ = 3
    on the hidden layers, per-split checkpoint directories holding the reference's keys, four floats back.  The same
    numpy / torch seeds give the same numbers; a manual composition of the same pieces for split 0 reproduces its error and
    MNLL exactly (the function IS that composition)."""
    from sklearn.model_selection import train_test_split
    from sklearn.preprocessing import StandardScaler
    from torch.utils.data import DataLoader, TensorDataset
    from whvi_amd.activations import Cosine
    from whvi_amd.evaluation import DeviceBatches, evaluate_bayesian_regression_dnn, make_optimizer
    rng = np.random.default_rng(3)
    X = rng.normal(size=(50, 5)).astype(np.float32) * 4 + 2
    y = (X[:, :2].sum(axis=1, keepdims=True) + 0.1 * rng.normal(size=(50, 1))).astype(np.float32)
    kwargs = dict(epochs1=1, epochs2=2, n_splits=2, batch_size=16, hidden=16, eval_samples=8, fast=False)

    def run(where):
        np.random.seed(0)
        torch.manual_seed(0)
        return evaluate_bayesian_regression_dnn(X, y, "cpu", where, **kwargs)
    first, second = run(tmp_path / "a"), run(tmp_path / "b")
    assert first == second and len(first) == 4 and all(np.isfinite(v) for v in first)
    out = capsys.readouterr().out
    assert "Iteration 1/2" in out and "Iteration 2/2" in out and out.count("Error:") == 4
    state = torch.load(tmp_path / "a" / "iter-1" / "epoch-0.pth")
    assert "likelihood.sigma" in state and "sequential.2.weight_submodule.s1" in state
    assert state["sequential.0.weight_submodule.weight_matrices.0.g_rho"].shape == (8,)       # (5 -> 16): two 8 x 8 blocks
    with pytest.raises(RuntimeError, match="needs a GPU"):
        evaluate_bayesian_regression_dnn(X, y, "cpu", tmp_path / "c", fast=True)
    # split 0 by hand
    np.random.seed(0)
    torch.manual_seed(0)
    Xs = StandardScaler().fit_transform(X)
    X_train, X_test, y_train, y_test = train_test_split(Xs, y, train_size=0.9, test_size=0.1)
    loader = DataLoader(TensorDataset(torch.tensor(X_train), torch.tensor(y_train)), batch_size=16)
    model = WHVIRegression([WHVILinear(5, 16, lambda_=3.0), nn.ReLU(), WHVILinear(16, 16, lambda_=3.0), nn.ReLU(),
                            WHVILinear(16, 1)], eval_samples=8)
    optimizer, scheduler = make_optimizer(model)
    model.train_model(loader, optimizer, scheduler, epochs1=1, epochs2=2, pbar_update_period=1)
    error0, mnll0 = model.eval_model(torch.tensor(X_test), torch.tensor(y_test))
    line = [ln for ln in out.splitlines() if ln.startswith("Error:")][0]
    assert line == f"Error: {error0}, MNLL: {mnll0}"
    # DeviceBatches == the DataLoader's batches (order, ragged tail, len(dataset))
    Xt, yt = torch.tensor(X_train), torch.tensor(y_train)
    views = DeviceBatches(Xt, yt, batch_size=16)
    assert len(views) == len(loader) == 3 and len(views.dataset) == len(loader.dataset) == 45
    for (a, b), (c, d) in zip(views, loader):
        assert torch.equal(a, c) and torch.equal(b, d)
    torch.manual_seed(9)                                                   # one pass consumes the host generator alike
    list(views)
    after_views = torch.rand(3)
    torch.manual_seed(9)
    list(loader)
    assert torch.equal(after_views, torch.rand(3))
    assert [tuple(a.shape) for a, _ in views] == [(16, 5), (16, 5), (13, 5)]
    with pytest.raises(ValueError):
        DeviceBatches(Xt, yt[:-1])
    assert torch.equal(Cosine()(Xt), torch.cos(Xt))


@pytest.mark.parametrize("D", [8, 64])
def test_diagonal_route_as_torch_ops_on_the_host(D, monkeypatch):
    """The square layer's route switch on HOST tensors: "auto" (the default) keeps the reference's dataflow there, ``True``
    applies the diagonal as torch ops (``DiagApplyFunction._reference_ops`` is the same expression), ``faithful_dataflow``
    round-trips.  The two agree to the dense-H path's rounding noise (src/weights.py:38-39 leaves ~1e-7 off-diagonals), for
    forward / forward_mc / direct sampling and every gradient; the closed-form backward of DiagApplyFunction (used with
    create_graph on the GPU) equals autograd through the same expression."""
    from whvi_amd.weights import DiagApplyFunction, WHVISquarePow2Matrix
    torch.manual_seed(D)
    sq = WHVISquarePow2Matrix(D, bias=True)
    with torch.no_grad():
        sq.s1.mul_(30.0), sq.s2.mul_(30.0), sq.g_mu.normal_(), sq.bias.normal_()
    assert sq._diag_mode() == "auto" and sq._diag_route(torch.zeros(2, D)) is None and not sq.faithful_dataflow
    x = torch.randn(9, D, requires_grad=True)
    results = {}
    for mode in (False, True):
        sq.exploit_diagonal = mode
        assert sq.faithful_dataflow == (mode is False) and sq._diag_route(x) == ("ops" if mode else None)
        outs = []
        for fn in (lambda: sq(x), lambda: sq.forward_mc(x, 3), lambda: sq.forward_mc(x, 3, relu_in=True, relu_out=True)):
            torch.manual_seed(1)
            sq.zero_grad(set_to_none=True)
            x.grad = None
            y = fn()
            (y.square().sum() + sq.kl).backward()
            outs.append([y.detach(), x.grad.clone()] + [p.grad.clone() for p in sq.parameters()])
        results[mode] = outs
    for a, b in zip(results[True], results[False]):
        for u, v in zip(a, b):
            assert u.shape == v.shape and float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 1e-12
    sq.faithful_dataflow = False
    assert sq.exploit_diagonal == "auto"
    # DiagApplyFunction's create_graph backward (pure torch) against autograd through _reference_ops, float64
    g = torch.Generator().manual_seed(3)
    S, B = 3, 5
    leaves = [torch.randn(S, B, D, generator=g, dtype=torch.float64), torch.randn(D, generator=g, dtype=torch.float64),
              torch.randn(D, generator=g, dtype=torch.float64), torch.randn(1 + S, D, generator=g, dtype=torch.float64),
              torch.randn(1, D, generator=g, dtype=torch.float64)]
    gout = torch.randn(S, B, D, generator=g, dtype=torch.float64)
    for relu_in, relu_out in ((False, False), (True, True)):
        ins = [t.clone().requires_grad_(True) for t in leaves]
        want = torch.autograd.grad(DiagApplyFunction._reference_ops(*ins, True, relu_in, relu_out), ins, gout)

        class Ctx:
            saved_tensors = (leaves[0], leaves[1], leaves[2], leaves[3], leaves[4] if relu_out else None)
            n_samples, mean_plus, bias_shape, needs_input_grad = S, True, (1, D), (True,) * 9
        Ctx.relu_in, Ctx.relu_out = relu_in, relu_out
        with torch.enable_grad():
            got = DiagApplyFunction.backward(Ctx, gout)
        for u, v in zip(got[:5], want):
            assert float((u - v).abs().max()) <= 1e-12 * float(v.abs().max()), (relu_in, relu_out)
