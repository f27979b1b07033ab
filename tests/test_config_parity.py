"""Parity at the BASELINE configurations' REAL shapes.

Golden data: tests/golden/config4_golden.npz and train_golden.npz, recorded from the live reference by
tests/golden/make_golden_r2.py (eps draws recorded, replayed here on any device).

* config 4 -- ``WHVILinear(3, 1024)`` (256 sub-matrices of D = 4; src/weights.py:135-160, :179-180),
  ``WHVILinear(1024, 1024)`` (src/weights.py:87-93), ``WHVILinear(1024, 1)`` (src/weights.py:239-248,
  src/layers.py:31-38) and the 3 -> 1024 -> 1024 -> 1 ``WHVIRegression`` (src/networks.py:47-51, :118-128):
  forward / KL / every gradient at 1e-5 relative (north_star), in the reference's per-sample loop AND in the batched
  Monte-Carlo pass; on the host and on the GPU.
* F4 -- ``make_optimizer`` (src/evaluation.py:15-27) and ``train_model`` (src/networks.py:71-99): the recorded
  trajectories (per-step loss / learning rate, final state, the epoch-0 checkpoint) replayed.
* configs 2, 3, 5 at their full sizes on the GPU: sampled rows against the oracle.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

from whvi_amd.evaluation import make_optimizer
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

from test_host import ReplayRandn

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _npz(name):
    return np.load(os.path.join(GOLD, name))


def _load_flat(module, names, flat):
    """Fill ``module``'s parameters from the reference's flat parameter vector; names must match one for one."""
    ours = [n for n, _ in module.named_parameters()]
    assert ours == str(names).split("\n"), "named_parameters() must match the reference name for name, in order"
    off = 0
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.from_numpy(flat[off:off + p.numel()]).view_as(p))
            off += p.numel()
    assert off == len(flat)


def _flat_grads(module):
    return torch.cat([p.grad.reshape(-1) for p in module.parameters()]).detach().cpu().numpy()


def _rel(got, want):
    want = np.asarray(want, dtype=np.float64)
    return float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / (np.abs(want).max() or 1.0))


def _grad_groups(module):
    """(label, slice) per parameter KIND of every WHVI layer: all ``s1`` of a stacked layer form one group (their
    gradients share a scale), so does each of ``s2 / g_mu / g_rho``; anything else is its own group."""
    groups, off = {}, 0
    for name, p in module.named_parameters():
        parts = name.split(".")
        kind = parts[-1]
        layer = ".".join(parts[:2]) if parts[0] == "sequential" else parts[0]
        groups.setdefault((layer, kind), []).append((off, off + p.numel()))
        off += p.numel()
    return groups


def _check_grads(module, want_flat, rtol, what):
    got = _flat_grads(module)
    for (layer, kind), spans in _grad_groups(module).items():
        idx = np.concatenate([np.arange(a, b) for a, b in spans])
        err = _rel(got[idx], want_flat[idx])
        assert err <= rtol, f"{what}: grad of {layer} {kind}: {err:.3e} > {rtol}"


class BatchedReplay:
    """``torch.randn`` for the batched Monte-Carlo pass: the k-th draw belongs to the k-th WHVI layer and has shape
    ``(J, S, D)``; ``tables[k]`` holds the reference's sequential draws as ``(S, J, D)``."""

    def __init__(self, tables):
        self.tables, self.i, self.real = list(tables), 0, torch.randn

    def __call__(self, *a, **k):
        size = tuple(a[0]) if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else tuple(a)
        if self.i >= len(self.tables):
            return self.real(*a, **k)
        t = torch.from_numpy(np.ascontiguousarray(np.swapaxes(self.tables[self.i], 0, 1)))
        self.i += 1
        assert tuple(t.shape) == size, (tuple(t.shape), size)
        dev = k.get("device", None)
        return t.to(dev) if dev is not None else t


# ---- config 4: the three layer shapes -----------------------------------------------------------------------
C4_LAYERS = ["st3x1024", "sq1024", "col1024x1"]


def run_config4_layer(name, device, monkeypatch, mode, rtol=1e-5):
    g = _npz("config4_golden.npz")
    b = {k.split("/", 2)[2]: g[k] for k in g.files if k.startswith(f"layer/{name}/")}
    layer = WHVILinear(int(b["n_in"]), int(b["n_out"]), lambda_=float(b["lambda_"]))
    _load_flat(layer, b["param_names"], b["params"])
    layer = layer.to(device)
    x = torch.from_numpy(b["x"]).to(device).requires_grad_(True)
    weight = torch.from_numpy(b["weight"]).to(device)
    eps = b["eps"]                                              # (draws, D) in the reference's draw order
    if mode == "loop":
        replay = ReplayRandn(list(eps))
        monkeypatch.setattr(torch, "randn", replay)
        y = layer(x)
        monkeypatch.undo()
        assert replay.i == len(eps)
    else:                                                       # one sample through the batched pass
        monkeypatch.setattr(torch, "randn", BatchedReplay([eps[None]]))
        y = layer.forward_mc(x, 1)[0]
        monkeypatch.undo()
    kl = layer.kl
    ((y * weight).sum() + kl).backward()
    assert _rel(y.detach().cpu().numpy(), b["y"]) <= rtol, f"{name} {mode}: forward"
    assert abs(float(kl) - float(b["kl"])) <= rtol * abs(float(b["kl"])), f"{name} {mode}: kl"
    assert _rel(x.grad.cpu().numpy(), b["grad_x"]) <= rtol, f"{name} {mode}: grad_x"
    _check_grads(layer, b["grads"], rtol, f"{name} {mode}")


@pytest.mark.parametrize("mode", ["loop", "batched"])
@pytest.mark.parametrize("name", C4_LAYERS)
def test_config4_layer_vs_reference_cpu(name, mode, monkeypatch):
    run_config4_layer(name, "cpu", monkeypatch, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["loop", "batched"])
@pytest.mark.parametrize("name", C4_LAYERS)
def test_config4_layer_vs_reference_gpu(name, mode, monkeypatch, hip_lib, dataflow):
    run_config4_layer(name, "cuda", monkeypatch, mode)


# ---- config 4: the network ----------------------------------------------------------------------------------
def _config4_net():
    return WHVIRegression([WHVILinear(3, 1024, lambda_=2.0), nn.ReLU(), WHVILinear(1024, 1024, lambda_=2.0), nn.ReLU(),
                           WHVILinear(1024, 1, lambda_=2.0)], train_samples=3, eval_samples=2)


def _replay_for(tables, mode):
    """randn replacement serving ``tables[k]`` = (S, J, D) of WHVI layer k in the order ``mode`` draws them."""
    if mode == "batched":
        return BatchedReplay(tables)
    S = tables[0].shape[0]
    flat = [t[s, j] for s in range(S) for t in tables for j in range(t.shape[1])]     # sample-major, layer, sub-matrix
    return ReplayRandn(flat)


def run_config4_network(device, monkeypatch, mode, rtol=1e-5):
    g = _npz("config4_golden.npz")
    net = _config4_net()
    _load_flat(net, g["net/param_names"], g["net/params"])
    net = net.to(device).train()
    net.mc_mode = mode
    tables = [g[f"net/eps_layer{k}"] for k in range(3)]
    x, y = torch.from_numpy(g["net/x"]).to(device), torch.from_numpy(g["net/y"]).to(device)
    monkeypatch.setattr(torch, "randn", _replay_for(tables, mode))
    loss = net.loss(x, y, n=100)
    monkeypatch.undo()
    loss.backward()
    assert abs(float(loss) - float(g["net/loss"])) <= rtol * abs(float(g["net/loss"]))
    assert abs(float(net.current_mnll) - float(g["net/mnll"])) <= rtol * abs(float(g["net/mnll"]))
    assert abs(float(net.current_kl) - float(g["net/kl"])) <= rtol * abs(float(g["net/kl"]))
    _check_grads(net, g["net/grads"], rtol, f"network {mode}")
    monkeypatch.setattr(torch, "randn", _replay_for(tables, mode))
    with torch.no_grad():
        pred = net(x)
    monkeypatch.undo()
    assert pred.shape == (6, 1, 3)
    assert _rel(pred.cpu().numpy(), g["net/pred"]) <= rtol, f"network {mode}: predictions"


@pytest.mark.parametrize("mode", ["loop", "batched"])
def test_config4_network_vs_reference_cpu(mode, monkeypatch):
    run_config4_network("cpu", monkeypatch, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["loop", "batched"])
def test_config4_network_vs_reference_gpu(mode, monkeypatch, hip_lib, dataflow):
    run_config4_network("cuda", monkeypatch, mode)


# ---- F4: make_optimizer + train_model trajectories --------------------------------------------------------------
def test_make_optimizer_is_the_reference_schedule():
    """src/evaluation.py:15-27: Adam(lr = lambda0) under LambdaLR(lambda0 * (1 + gamma t)^-p) -> lambda0^2 * ..."""
    g = _npz("train_golden.npz")
    for run, kwargs in (("default", {}), ("fast", {"lambda0": 0.05})):
        net = nn.Linear(2, 2)
        opt, sched = make_optimizer(net, **kwargs)
        assert isinstance(opt, torch.optim.Adam) and isinstance(sched, torch.optim.lr_scheduler.LambdaLR)
        lrs = []
        for _ in range(len(g[f"{run}/lr"])):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
        assert np.array_equal(np.array(lrs), g[f"{run}/lr"]), run
    lam0 = 0.001
    assert make_optimizer(nn.Linear(2, 2))[0].param_groups[0]["lr"] == lam0 * lam0     # the documented quirk


def run_train_trajectory(run, device, monkeypatch, tmp_path, mode, loss_rtol_first, loss_rtol, state_rtol):
    g = _npz("train_golden.npz")
    S = 2
    net = WHVIRegression([WHVILinear(3, 16, lambda_=3.0), nn.ReLU(), WHVILinear(16, 16, lambda_=3.0), nn.ReLU(),
                          WHVILinear(16, 1, lambda_=3.0)], train_samples=S, eval_samples=4)
    init = {k[len(f"{run}/init."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{run}/init.")}
    assert set(init) == set(net.state_dict()), "state_dict keys must interchange with the reference"
    net.load_state_dict(init)
    net = net.to(device)
    net.mc_mode = mode
    X, Y = torch.from_numpy(g[f"{run}/X"]).to(device), torch.from_numpy(g[f"{run}/Y"]).to(device)
    loader = DataLoader(TensorDataset(X, Y), batch_size=8)
    import ast
    optimizer, scheduler = make_optimizer(net, **ast.literal_eval(str(g[f"{run}/optimizer_kwargs"])))
    epochs1, epochs2 = (int(v) for v in g[f"{run}/epochs"])
    steps = len(g[f"{run}/loss"])
    tables = [g[f"{run}/eps_layer{k}"] for k in range(3)]                 # (steps, S, J, D)
    if mode == "batched":
        draws = BatchedReplay([t[i] for i in range(steps) for t in tables])
    else:
        draws = ReplayRandn([t[i, s, j] for i in range(steps) for s in range(S) for t in tables
                             for j in range(t.shape[2])])
    trace = {"loss": [], "lr": []}
    inner = net.loss

    def traced(*a, **k):
        trace["lr"].append(optimizer.param_groups[0]["lr"])
        value = inner(*a, **k)
        trace["loss"].append(float(value))
        return value
    net.loss = traced
    monkeypatch.setattr(torch, "randn", draws)
    net.train_model(loader, optimizer, scheduler, epochs1=epochs1, epochs2=epochs2, checkpoint_dir=tmp_path)
    monkeypatch.undo()
    assert draws.i == len(draws.draws if mode == "loop" else draws.tables), "every recorded eps consumed, no more"
    assert not net.training and int(g[f"{run}/training_flag_after"]) == 0     # train_model ends in eval mode
    assert np.array_equal(np.array(trace["lr"]), g[f"{run}/lr"]), "learning-rate schedule"
    want = g[f"{run}/loss"]
    got = np.array(trace["loss"])
    assert abs(got[0] - want[0]) <= loss_rtol_first * abs(want[0]), (got[0], want[0])
    assert np.abs(got - want).max() <= loss_rtol * np.abs(want).max(), np.abs(got - want).max()

    def close_state(state, prefix):
        keys = [k[len(prefix):] for k in g.files if k.startswith(prefix)]
        assert set(keys) == set(state)
        for k in keys:
            ref, start = g[prefix + k].astype(np.float64), g[f"{run}/init.{k}"].astype(np.float64)
            ours = state[k].detach().cpu().numpy().astype(np.float64)
            # error relative to how far training moves a parameter: Adam steps are ~lr per step whatever the gradient's
            # size, so the yardstick is max(actual movement, sum of learning rates) -- entries whose gradient is pure
            # rounding noise (most of this toy net: the as-written column layer reads hidden unit 0 only) wander by
            # noise-sized steps -- floored at fp32 resolution of the values themselves
            moved = max(np.abs(ref - start).max(), float(np.sum(g[f"{run}/lr"])))
            tol = state_rtol * moved + 4e-7 * np.abs(ref).max()
            assert np.abs(ours - ref).max() <= tol, (prefix + k, np.abs(ours - ref).max(), moved)
    close_state(net.state_dict(), f"{run}/final.")
    # the checkpoint written inside phase 2 (src/networks.py:95-96): same file name, same keys, same values
    assert sorted(os.listdir(tmp_path)) == ["epoch-0.pth"]
    close_state(torch.load(tmp_path / "epoch-0.pth"), f"{run}/ckpt_epoch0.")


@pytest.mark.parametrize("run", ["default", "fast"])
def test_train_trajectory_vs_reference_cpu(run, monkeypatch, tmp_path):
    # host path = the reference's own torch ops in the same order: far inside the GPU bars below
    run_train_trajectory(run, "cpu", monkeypatch, tmp_path, "loop", 1e-6, 1e-6, 1e-3)


def test_train_trajectory_batched_pass_cpu(monkeypatch, tmp_path):
    run_train_trajectory("fast", "cpu", monkeypatch, tmp_path, "batched", 1e-5, 1e-5, 2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["loop", "batched"])
@pytest.mark.parametrize("run", ["default", "fast"])
def test_train_trajectory_vs_reference_gpu(run, mode, monkeypatch, tmp_path, hip_lib, dataflow):
    """Bars: 1e-5 relative on the first step's loss (north_star) and on all 21; parameters after 21 Adam steps
    within 2 % of the distance the reference moved them (Adam normalises gradients, so rounding-level gradient
    differences on near-zero entries turn into O(lr) step differences; the looser bound states that)."""
    run_train_trajectory(run, "cuda", monkeypatch, tmp_path, mode, 1e-5, 1e-5, 2e-2)


def test_device_resident_schedule_is_the_reference_schedule():
    """``make_optimizer(capturable=True)``'s ``DeviceLambdaLR`` (state in tensors, ``step()`` = device ops only) follows the
    recorded learning rates of the reference's ``LambdaLR`` (src/evaluation.py:25-26) to float32 resolution -- here on
    host tensors; the GPU trajectory test below runs it inside the captured step."""
    from whvi_amd.evaluation import DeviceLambdaLR
    g = _npz("train_golden.npz")
    for run, lam0 in (("default", 0.001), ("fast", 0.05)):
        net = nn.Linear(2, 2)
        opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(lam0), foreach=False)
        sched = DeviceLambdaLR(opt, lambda t: lam0 * torch.pow(1.0 + 0.0005 * t, -0.3), base_lrs=[lam0])
        want = g[f"{run}/lr"]
        for i in range(len(want)):
            assert torch.is_tensor(opt.param_groups[0]["lr"])
            assert abs(sched.get_last_lr()[0] - want[i]) <= 2e-7 * want[i], (run, i)
            sched.step()
        state = sched.state_dict()
        sched.load_state_dict({"t": 3.0, "base_lrs": state["base_lrs"]})
        assert abs(sched.get_last_lr()[0] - want[3]) <= 2e-7 * want[3]
    host = torch.optim.lr_scheduler.LambdaLR(torch.optim.Adam(nn.Linear(2, 2).parameters(), lr=1.0), lambda t: 1.0)
    from whvi_amd.graphs import GraphedTrainStep
    assert not getattr(host, "device_resident", False) and DeviceLambdaLR.device_resident and GraphedTrainStep is not None


@pytest.mark.gpu
def test_one_launch_schedule_is_the_reference_schedule_gpu(hip_lib):
    """``DeviceDecayLR`` (what ``make_optimizer(capturable=True)`` returns: ``whvi_decay_lr_step``, one single-thread launch
    per step) against the reference's host-side ``LambdaLR`` (src/evaluation.py:25-26) over 3 000 steps and against the
    recorded rates: float32 resolution, the step counter exact, two parameter groups advance together, state round trip,
    and the generic ``DeviceLambdaLR`` (tensor ops) agrees with it."""
    from whvi_amd.evaluation import DeviceDecayLR, DeviceLambdaLR, make_optimizer
    g = _npz("train_golden.npz")
    for run, lam0 in (("default", 0.001), ("fast", 0.05)):
        net = nn.Linear(2, 2).to("cuda")
        opt, sched = make_optimizer(net, lambda0=lam0, capturable=True)
        assert isinstance(sched, DeviceDecayLR) and opt.defaults.get("fused") is True
        want = g[f"{run}/lr"]
        for i in range(len(want)):
            assert abs(sched.get_last_lr()[0] - want[i]) <= 2e-7 * want[i], (run, i)
            sched.step()
        assert float(sched.t) == len(want)
    a, b = nn.Linear(2, 2).to("cuda"), nn.Linear(3, 3).to("cuda")
    opt = torch.optim.Adam([{"params": a.parameters(), "lr": 0.01}, {"params": b.parameters(), "lr": 0.002}], capturable=True)
    sched = DeviceDecayLR(opt, 0.001, 0.0005, 0.3)
    twin = torch.optim.Adam([{"params": a.parameters(), "lr": 0.01}, {"params": b.parameters(), "lr": 0.002}], capturable=True)
    generic = DeviceLambdaLR(twin, lambda t: 0.001 * torch.pow(1.0 + 0.0005 * t, -0.3))
    host_opt = torch.optim.Adam([{"params": nn.Linear(2, 2).parameters(), "lr": 0.01},
                                 {"params": nn.Linear(3, 3).parameters(), "lr": 0.002}])
    host = torch.optim.lr_scheduler.LambdaLR(host_opt, lambda t: 0.001 * ((1 + 0.0005 * t) ** (-0.3)))
    for i in range(3000):
        if i % 250 == 0 or i < 8:
            for got, other, ref in zip(sched.get_last_lr(), generic.get_last_lr(), host.get_last_lr()):
                assert abs(got - ref) <= 1.2e-7 * ref and abs(other - ref) <= 1.2e-7 * ref, (i, got, other, ref)
        sched.step()
        generic.step()
        host_opt.step()
        host.step()
    assert float(sched.t) == 3000.0
    state = sched.state_dict()
    sched.load_state_dict({"t": 7.0, "base_lrs": state["base_lrs"]})
    assert abs(sched.get_last_lr()[1] - 0.002 * 0.001 * (1 + 0.0005 * 7) ** -0.3) <= 1.2e-7 * 2e-6
    from whvi_amd import _hip
    with pytest.raises(RuntimeError):
        _hip.decay_lr_step(torch.zeros((), device="cuda"), torch.zeros((), device="cuda"), 1.0, 1.0, 1.0, 1.0)   # t not float64


@pytest.fixture
def warn_always():
    """torch's C++ warnings fire once per process unless warn-always is on: a test that promotes one to an error must not
    depend on no earlier test having used up the single emission."""
    before = torch.is_warn_always_enabled()
    torch.set_warn_always(True)
    yield
    torch.set_warn_always(before)


@pytest.mark.gpu
@pytest.mark.filterwarnings("error:The AccumulateGrad node's stream does not match:UserWarning")
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("run", ["default", "fast"])
def test_train_trajectory_graphed_vs_reference_gpu(run, packed, tmp_path, hip_lib, dataflow, warn_always):
    """The reference's training recipe on the FAST path: ``train_model(graphed=True)`` -- both phases,
    ``scheduler.step()`` after every batch (src/networks.py:80-81), the checkpoint of phase 2 -- with every step one
    hipGraph replay holding loss, backward, Adam and the schedule (learning rate and step counter in device memory).
    The two recorded runs of the reference's ``train_model`` + ``make_optimizer`` are replayed through it with the recorded
    eps injected into the layers' static buffers: learning rates equal to float32 resolution, every loss within 1e-5,
    final state and ``epoch-0.pth`` inside the 2 % movement bound of the eager GPU test, reference ``state_dict`` keys --
    also with the packed parameter layout."""
    import ast
    g = _npz("train_golden.npz")
    S = 2
    net = WHVIRegression([WHVILinear(3, 16, lambda_=3.0), nn.ReLU(), WHVILinear(16, 16, lambda_=3.0), nn.ReLU(),
                          WHVILinear(16, 1, lambda_=3.0)], train_samples=S, eval_samples=4)
    init = {k[len(f"{run}/init."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{run}/init.")}
    net.load_state_dict(init)
    net = net.to("cuda")
    X, Y = torch.from_numpy(g[f"{run}/X"]).to("cuda"), torch.from_numpy(g[f"{run}/Y"]).to("cuda")
    loader = DataLoader(TensorDataset(X, Y), batch_size=8)
    optimizer, scheduler = make_optimizer(net, **ast.literal_eval(str(g[f"{run}/optimizer_kwargs"])), capturable=True,
                                          packed=packed)
    assert len(list(net.parameters())) == (13 if packed else 25)
    epochs1, epochs2 = (int(v) for v in g[f"{run}/epochs"])
    steps = len(g[f"{run}/loss"])
    tables = [g[f"{run}/eps_layer{k}"] for k in range(3)]                 # (steps, S, J, D)
    seen = {"i": 0, "lr": [], "loss": []}

    def before_replay(step):
        i = seen["i"]
        if i > 0:
            seen["loss"].append(float(step.static_loss))                  # the previous replay's loss
        assert len(step.eps_buffers) == 3
        for buf, table in zip(step.eps_buffers, tables):
            buf.copy_(torch.from_numpy(np.ascontiguousarray(np.swapaxes(table[i], 0, 1))))
        seen["lr"].append(float(optimizer.param_groups[0]["lr"]))
        seen["i"] = i + 1
    before = {k: v.clone() for k, v in net.state_dict().items()}
    step = net.train_model(loader, optimizer, scheduler, epochs1=epochs1, epochs2=epochs2, checkpoint_dir=tmp_path,
                           graphed=True, packed=packed, graph_options={"static_eps": True, "before_replay": before_replay})
    seen["loss"].append(float(step.static_loss))
    assert seen["i"] == steps and not net.training
    assert float(scheduler.t) == steps, "the schedule advanced once per replay and never during the capture warm-up"
    assert all(torch.equal(init[k].to("cuda"), before[k]) for k in init)
    want_lr = g[f"{run}/lr"]
    assert np.abs(np.array(seen["lr"]) - want_lr).max() <= 2e-7 * want_lr.max(), "learning-rate schedule"
    want, got = g[f"{run}/loss"], np.array(seen["loss"])
    assert abs(got[0] - want[0]) <= 1e-5 * abs(want[0]), (got[0], want[0])
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), np.abs(got - want).max()

    def close_state(state, prefix):
        keys = [k[len(prefix):] for k in g.files if k.startswith(prefix)]
        assert set(keys) == set(state), "the reference's state_dict keys"
        for k in keys:
            ref, start = g[prefix + k].astype(np.float64), g[f"{run}/init.{k}"].astype(np.float64)
            ours = state[k].detach().cpu().numpy().astype(np.float64)
            moved = max(np.abs(ref - start).max(), float(np.sum(g[f"{run}/lr"])))
            assert np.abs(ours - ref).max() <= 2e-2 * moved + 4e-7 * np.abs(ref).max(), (prefix + k, np.abs(ours - ref).max(), moved)
    close_state(net.state_dict(), f"{run}/final.")
    assert sorted(os.listdir(tmp_path)) == ["epoch-0.pth"]
    close_state(torch.load(tmp_path / "epoch-0.pth"), f"{run}/ckpt_epoch0.")
    # the layers have their generator back: a later eager pass draws fresh eps
    assert all(getattr(m, "_eps_static", None) is None for m in net.modules())


# ---- configs 2, 3, 5 at their full sizes (GPU) ----------------------------------------------------------------
def _bits(a):
    return a.view({2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


@pytest.mark.gpu
def test_config2_full_size_forward_mc_vs_oracle(monkeypatch, hip_lib, dataflow):
    """BASELINE config 2 as timed: ``WHVILinear(512, 512)`` forward + KL, 32 MC samples, batch 4096, fp32, through
    ``forward_mc``.  Every sample's output on 48 sampled batch rows against oracle/whvi_oracle.py (pinned to the
    reference's bundles), 1e-5 relative; KL against the oracle's."""
    from oracle import whvi_oracle as wo
    D, S, B = 512, 32, 4096
    torch.manual_seed(12)
    layer = WHVILinear(D, D, lambda_=0.7)
    with torch.no_grad():
        sq = layer.weight_submodule
        sq.g_mu.copy_(torch.randn(D) * 0.3)
        sq.s1.mul_(10.0)
        sq.s2.mul_(10.0)
    params = {k: v.detach().numpy().copy() for k, v in layer.named_parameters()}
    rng = np.random.default_rng(5)
    eps = rng.standard_normal((S, 1, D)).astype(np.float32)
    x = rng.standard_normal((B, D)).astype(np.float32)
    layer = layer.to("cuda")
    monkeypatch.setattr(torch, "randn", BatchedReplay([eps]))
    with torch.no_grad():
        out = layer.forward_mc(torch.from_numpy(x).to("cuda"), S)               # (S, B, D)
    monkeypatch.undo()
    assert out.shape == (S, B, D)
    rows = np.unique(np.concatenate([[0, 1, B - 1], rng.integers(0, B, 45)]))
    got = out[:, torch.from_numpy(rows).to("cuda")].cpu().numpy()
    ref = wo.layer_from_params(D, D, 0.7, params)
    scale = 0.0
    want = np.stack([ref.forward(x[rows], eps[k, 0]) for k in range(S)])
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 1e-5 * scale
    assert abs(float(layer.kl) - float(ref.kl)) <= 1e-5 * abs(float(ref.kl))
    assert bool(torch.isfinite(out).all())


@pytest.mark.gpu
@pytest.mark.parametrize("in_place", [False, True])
def test_config3_full_size_fused_vs_oracle(in_place, hip_lib):
    """BASELINE config 3 as timed (bench.py ``_extra_fused``): the fused S.H.diag(g).H.S kernel, D = 2048, 64 MC
    samples, batch 8192 = 2^19 rows (4 GiB), column axis, rows in (batch, sample) order -- the production
    instantiation (streaming launch, LDS-staged a / c).  96 sampled rows bit-exact vs ``oracle.pipeline``; every
    other row checked for being written (finite, non-zero)."""
    import oracle
    from whvi_amd import _hip
    free, _ = torch.cuda.mem_get_info()
    if free < 10 * 2 ** 30:
        pytest.skip("needs ~9 GiB of free HBM")
    d, S, B = 2048, 64, 8192
    rows = B * S
    g = torch.Generator(device="cuda").manual_seed(33)
    x = torch.randn(rows, d, device="cuda", generator=g)
    a, c = torch.randn(d, device="cuda", generator=g) * 0.1, torch.randn(d, device="cuda", generator=g) * 0.1
    b = torch.randn(S, d, device="cuda", generator=g)
    rng = np.random.default_rng(3)
    idx = np.unique(np.concatenate([[0, 1, 63, 64, rows // 2 + 5, rows - 65, rows - 1], rng.integers(0, rows, 89)]))
    tidx = torch.from_numpy(idx).to("cuda")
    src_rows = x[tidx].cpu().numpy()
    out = _hip.fused_shs(x, a, b, c, axis="col", n_samples=S, sample_stride=1, out=x if in_place else None)
    assert (out.data_ptr() == x.data_ptr()) == in_place
    got = out[tidx].cpu().numpy()
    # every gathered row is its own "sample": its g vector is b[row % S]
    want = oracle.pipeline(src_rows, a.cpu().numpy(), b.cpu().numpy()[idx % S], c.cpu().numpy(),
                           n_samples=len(idx), sample_stride=1, axis="col")
    assert np.array_equal(_bits(got), _bits(want))
    nz = (out != 0).any(dim=1)
    assert bool(nz.all()) and bool(torch.isfinite(out[:: 4099]).all())
    if not in_place:
        assert np.array_equal(x[tidx].cpu().numpy(), src_rows), "out-of-place must leave the source untouched"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_config5_full_size_16bit_fwht_vs_oracle(dtype, hip_lib):
    """BASELINE config 5 (one GPU's share): D = 4096, 2^20 rows of fp16 (8 GiB), in place, the streaming launch with
    the LDS-staged network.  Contract: f32 arithmetic, one RNE rounding on store -> sampled rows bit-equal to
    ``oracle(x.float()).to(dtype)``; small-integer data additionally satisfies H.H = D.I exactly in fp16."""
    import oracle
    from whvi_amd import _hip
    free, _ = torch.cuda.mem_get_info()
    if free < 20 * 2 ** 30:
        pytest.skip("needs ~17 GiB of free HBM")
    d, rows = 4096, 1 << 20
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.empty(rows, d, device="cuda", dtype=dtype)
    step = 1 << 17
    for r in range(0, rows, step):
        x[r:r + step] = (torch.randn(step, d, device="cuda", generator=g) * 0.25).to(dtype)
    rng = np.random.default_rng(9)
    idx = np.unique(np.concatenate([[0, 1, 4095, rows // 2 + 7, rows - 1], rng.integers(0, rows, 60)]))
    tidx = torch.from_numpy(idx).to("cuda")
    src = x[tidx].float().cpu().numpy()
    _hip.fwht_rows(x, out=x)
    got = x[tidx].cpu()
    want = torch.from_numpy(oracle.fwht(src)).to(dtype)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert bool(torch.isfinite(x[:: 4099].float()).all())
    # involution on exactly representable data: entries in {-1, 0, 1}, |H x| <= 4096 and H H x = 4096 x (|.| <= 4096)
    if dtype == torch.float16:
        for r in range(0, rows, step):
            x[r:r + step] = torch.randint(-1, 2, (step, d), generator=g, device="cuda", dtype=torch.int32).to(dtype)
        keep = x[tidx].clone()
        _hip.fwht_rows(x, out=x)
        _hip.fwht_rows(x, out=x)
        assert torch.equal(x[tidx], keep * 4096)


def _oracle_config4_predictions(params, x_rows, tables):
    """(rows, 1, S) predictions of the 3 -> 1024 -> 1024 -> 1 network from oracle/whvi_oracle.py -- the numpy restatement of
    src/weights.py / src/layers.py / src/networks.py:47-51 on the C butterfly oracle, pinned to the reference's recorded
    bundles in tests/test_oracle.py -- one Monte-Carlo sample at a time, like the reference's loop."""
    from oracle import whvi_oracle as wo

    def layer(index, n_in, n_out):
        prefix = f"sequential.{index}."
        return wo.layer_from_params(n_in, n_out, 2.0, {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)})
    first, middle, last = layer(0, 3, 1024), layer(2, 1024, 1024), layer(4, 1024, 1)
    S = tables[0].shape[0]
    preds = []
    for k in range(S):
        h = np.maximum(first.forward(x_rows, list(tables[0][k])), np.float32(0))
        h = np.maximum(middle.forward(h, tables[1][k, 0]), np.float32(0))
        preds.append(last.forward(h, tables[2][k, 0]))
    return np.stack(preds, axis=2).astype(np.float32)


def run_config4_real_size(device, monkeypatch):
    """The reference's own predictions for config 4's network at a real batch -- 512 rows x 16 MC samples, recorded from the live
    reference with every eps (tests/golden/make_golden_r4.py; src/networks.py:36-54) -- against ``forward_batched`` (one
    batched pass for all samples) at north_star's 1e-5, and ``eval_model``'s RMSE / MNLL of them (src/networks.py:101-133).

    The fixture holds the reference's output under BOTH of its device dispatches (src/weights.py:34-41): ``pred`` = the host
    route (dense H matmul for D < 4096) and ``pred_butterfly`` = the GPU route (butterfly FWHT for every matrix; the
    reference's own vectorised FWHT standing in for its CUDA kernel).  They differ by 2.8e-5 of the largest prediction (the
    dense H product leaves ~1e-7 off-diagonals in the exactly diagonal as-written weight); each device is compared with the
    record of ITS dispatch at 1e-5 and with the other one at 1e-4."""
    g = _npz("config4_real_size_golden.npz")
    mine, other = ("pred", "pred_butterfly") if device == "cpu" else ("pred_butterfly", "pred")
    tag = "" if device == "cpu" else "_butterfly"
    S = g["eps_layer0"].shape[0]
    net = _config4_net()
    net.eval_samples = S
    _load_flat(net, g["param_names"], g["params"])
    net = net.to(device).eval()
    tables = [g[f"eps_layer{k}"] for k in range(3)]
    x, y = torch.from_numpy(g["x"]).to(device), torch.from_numpy(g["y"]).to(device)
    monkeypatch.setattr(torch, "randn", BatchedReplay(tables))
    with torch.no_grad():
        pred = net.forward_batched(x, S)
    monkeypatch.undo()
    assert pred.shape == (512, 1, 16)
    assert _rel(pred.cpu().numpy(), g[mine]) <= 1e-5
    assert _rel(pred.cpu().numpy(), g[other]) <= 1e-4
    # ... which is far below what distinguishes one Monte-Carlo sample from another
    assert float(g["pred"].std(axis=2).mean()) > 100 * 1e-5 * float(np.abs(g["pred"]).max())
    net.mc_mode = "batched"
    monkeypatch.setattr(torch, "randn", BatchedReplay(tables))
    with torch.no_grad():
        rmse, mnll = net.eval_model(x, y)
    monkeypatch.undo()
    assert abs(rmse - float(g["rmse" + tag])) <= 1e-5 * float(g["rmse" + tag])
    assert abs(mnll - float(g["mnll" + tag])) <= 1e-5 * abs(float(g["mnll" + tag]))
    return net, tables, g


def test_config4_real_size_vs_reference_cpu(monkeypatch):
    net, tables, g = run_config4_real_size("cpu", monkeypatch)
    # the oracle used for the full-size GPU check below reproduces the reference's recorded predictions too (sampled rows)
    rows = np.array([0, 1, 255, 511])
    params = {k: v.detach().numpy() for k, v in net.named_parameters()}
    want = _oracle_config4_predictions(params, g["x"][rows], tables)
    assert np.abs(want - g["pred_butterfly"][rows]).max() <= 1e-5 * np.abs(g["pred_butterfly"]).max()      # the butterfly dispatch


@pytest.mark.gpu
def test_config4_real_size_vs_reference_gpu(monkeypatch, hip_lib, dataflow):
    run_config4_real_size("cuda", monkeypatch)


@pytest.mark.gpu
def test_config4_full_size_share_vs_oracle(monkeypatch, hip_lib, dataflow):
    """BASELINE config 4 at one GPU's share of its timed size -- the 3 -> 1024 -> 1024 -> 1 network, protein-sized batch
    (45 730 rows), 16 of the 128 MC samples, batched predictive pass on the GPU -- on 24 sampled batch rows x all samples
    against oracle/whvi_oracle.py (the numpy + C-butterfly restatement of the reference's layers, pinned to the reference's
    bundles and, on this very network, to its recorded real-size predictions: the CPU test above), 1e-5."""
    S, B = 16, 45730
    g = _npz("config4_golden.npz")
    net = _config4_net()
    _load_flat(net, g["net/param_names"], g["net/params"])          # the reference-recorded (perturbed) parameters
    net.eval()
    params = {k: v.detach().numpy().copy() for k, v in net.named_parameters()}
    dev = net.to("cuda")
    rng = np.random.default_rng(45730)
    x = rng.standard_normal((B, 3)).astype(np.float32)
    tables = [rng.standard_normal((S, 256, 4)).astype(np.float32), rng.standard_normal((S, 1, 1024)).astype(np.float32),
              rng.standard_normal((S, 1, 1024)).astype(np.float32)]
    monkeypatch.setattr(torch, "randn", BatchedReplay(tables))
    with torch.no_grad():
        pred = dev.forward_batched(torch.from_numpy(x).to("cuda"), S)             # (B, 1, S)
    monkeypatch.undo()
    assert pred.shape == (B, 1, S) and bool(torch.isfinite(pred).all())
    rows = np.unique(np.concatenate([[0, 1, B - 1], rng.integers(0, B, 21)]))
    want = _oracle_config4_predictions(params, x[rows], tables)
    got = pred[torch.from_numpy(rows).to("cuda")].cpu().numpy()
    assert _rel(got, want) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("mean", [False, True])
def test_wbar_backward_at_streaming_size(mean, hip_lib):
    """The backward of the weight construction at a size that takes the streaming launch (D = 2048 x 40 matrices =
    640 MiB of dL/dW: non-temporal loads, the signed DPP network): every output against the float64 closed form the
    as-written matrix implies (W = D diag(s1 u s2): dL/du_i = D s1_i s2_i gW_ii, ...) and bit-identical to the
    LDS-staged network run on two of the matrices alone."""
    from whvi_amd import _hip
    D, S = 2048, 40
    g = torch.Generator(device="cuda").manual_seed(7)
    s1, s2 = torch.randn(1, D, device="cuda", generator=g), torch.randn(1, D, device="cuda", generator=g)
    u = torch.randn(1, S + (1 if mean else 0), D, device="cuda", generator=g)
    gw = torch.randn(1, S, D, D, device="cuda", generator=g)
    out = _hip.wbar_bwd(gw, s1, u, s2, mean=mean)
    assert _hip.last_kernel() == f"whvi::wbar_bwd_kernel<float, 11, 16, true, {'true' if mean else 'false'}, 0>"
    first = 1 if mean else 0
    diag = torch.diagonal(gw, dim1=2, dim2=3).double()                         # (1, S, D)
    uk = u[:, first:].double()
    want_u = D * s1.double().unsqueeze(1) * s2.double().unsqueeze(1) * diag
    u_tot = uk + (u[:, :1].double() if mean else 0.0)
    want_s2 = D * s1.double().unsqueeze(1) * u_tot * diag
    want_s1 = D * u_tot * s2.double().unsqueeze(1) * diag
    noise = 1e-6 * D * float(gw.abs().max()) * 8 * float(s1.abs().max() * s2.abs().max() * u.abs().max())
    for got, want, name in ((out[0], want_u, "u"), (out[1], want_s1, "s1"), (out[2], want_s2, "s2")):
        assert float((got[:, first:].double() - want).abs().max()) <= noise, name
    k0 = 17
    sub_u = torch.cat((u[:, :1], u[:, first + k0:first + k0 + 2]), dim=1) if mean else u[:, k0:k0 + 2]
    gw2, u2 = gw[:, k0:k0 + 2].contiguous(), sub_u.contiguous()             # two of the matrices alone: 32 MiB
    lds = _hip.wbar_bwd(gw2, s1, u2, s2, mean=mean, tiles="big")
    assert _hip.last_kernel().endswith(", 16, false, " + ("true" if mean else "false") + ", 2>")      # LDS-staged network
    assert torch.equal(lds[:, :, first:], out[:, :, first + k0:first + k0 + 2])
    quarter = _hip.wbar_bwd(gw2, s1, u2, s2, mean=mean)                      # what a problem of this size takes by default
    assert "wbar_bwd_kernel<float, 11, 8, false, " in _hip.last_kernel()     # quarter-size tiles, DPP network
    assert torch.equal(quarter[:, :, first:], lds[:, :, first:])


@pytest.mark.gpu
@pytest.mark.parametrize("D,S", [(2048, 20), (512, 336), (4096, 6)])
def test_weight_construction_at_streaming_size(D, S, hip_lib):
    """The weight construction with the mean matrix added (the `forward_mc` form) at a size that takes the streaming
    launch -- non-temporal stores, XCD-sliced / matrix-fastest block order (> 256 MiB of matrices): bit-identical to the
    same matrices built one sample at a time (cache-resident launches, plain order), and equal to the closed form
    D diag(s1 (u_mean + u_k) s2) within fp32 rounding."""
    from whvi_amd import _hip
    g = torch.Generator(device="cuda").manual_seed(D + S)
    s1, s2 = torch.randn(1, D, device="cuda", generator=g), torch.randn(1, D, device="cuda", generator=g)
    u = torch.randn(1, 1 + S, D, device="cuda", generator=g)
    mean = _hip.wbar_fwd(s1, u, s2, D, first=0, count=1).view(1, D, D)
    full = _hip.wbar_fwd(s1, u, s2, D, base=mean, first=1)                       # (1, S, D, D)
    assert full.numel() * 4 > 256 << 20 and _hip.last_kernel().endswith(", 16, true>")
    for k in (0, 1, S // 2, S - 1):
        one = _hip.wbar_fwd(s1, u, s2, D, base=mean, first=1 + k, count=1)
        assert _hip.last_kernel().endswith(", false>")
        assert torch.equal(one[0, 0], full[0, k]), k
    diag = torch.diagonal(full[0], dim1=1, dim2=2).double().cpu()                # (S, D)
    want = D * s1[0].double().cpu() * (u[0, :1] + u[0, 1:]).double().cpu() * s2[0].double().cpu()
    assert float((diag - want).abs().max()) <= 4e-6 * float(want.abs().max())
    off = full[0, S - 1].clone()
    off.diagonal().zero_()
    assert float(off.abs().max()) == 0.0                                          # exactly diagonal (SURVEY finding 1)


@pytest.mark.gpu
@pytest.mark.parametrize("max_shapes", [1, 2])
def test_graphed_training_with_a_ragged_last_batch(max_shapes, hip_lib):
    """``train_model(graphed=True)`` on a data set whose last batch is short (20 rows in batches of 8: 8, 8, 4): the full
    batches are hipGraph replays; the short one is a second captured step sharing the optimizer state and the schedule
    (``max_shapes=2``, the default) or takes the eager step BESIDE the captured one (``max_shapes=1``) -- detached from the
    graph's static gradient buffers first, or its backward would accumulate into what the last replay left there.  Same
    generator seed -> the run equals the all-eager run of the same recipe (same draws in the same order), every step
    counted by the device-resident schedule.  A second ``train_model`` call continues on the same captured steps."""
    import copy
    torch.manual_seed(8)
    net = WHVIRegression([WHVILinear(3, 16, lambda_=2.0), nn.ReLU(), WHVILinear(16, 16, lambda_=2.0), nn.ReLU(),
                          WHVILinear(16, 1, lambda_=2.0)], train_samples=2).to("cuda")
    twin = copy.deepcopy(net)
    X, Y = torch.randn(20, 3, device="cuda"), torch.randn(20, 1, device="cuda")
    loader = DataLoader(TensorDataset(X, Y), batch_size=8)
    opt, sched = make_optimizer(net, lambda0=0.05, capturable=True)
    opt2, sched2 = make_optimizer(twin, lambda0=0.05, capturable=True)
    torch.manual_seed(77)
    step = net.train_model(loader, opt, sched, epochs1=2, epochs2=2, graphed=True, graph_options={"max_shapes": max_shapes})
    assert step is not None and float(sched.t) == 12.0                     # 4 epochs x 3 batches, replayed or eager
    assert len(net._train_graphs["steps"]) == max_shapes and tuple(step.static_x.shape) == (8, 3)
    torch.manual_seed(77)
    twin.train_model(loader, opt2, sched2, epochs1=2, epochs2=2)           # the same recipe, all eager
    assert float(sched2.t) == 12.0 and abs(sched.get_last_lr()[0] - sched2.get_last_lr()[0]) < 1e-12
    for (k, a), (_, b) in zip(net.named_parameters(), twin.named_parameters()):
        assert bool(torch.isfinite(a).all()) and torch.allclose(a, b, rtol=1e-4, atol=1e-6), k
    # a later call with the same optimizer / schedule / loader replays the SAME captured steps (no re-capture)
    torch.manual_seed(78)
    again = net.train_model(loader, opt, sched, epochs1=0, epochs2=1, graphed=True, graph_options={"max_shapes": max_shapes})
    assert again is step and len(net._train_graphs["steps"]) == max_shapes and float(sched.t) == 15.0
    torch.manual_seed(78)
    twin.train_model(loader, opt2, sched2, epochs1=0, epochs2=1)
    for (k, a), (_, b) in zip(net.named_parameters(), twin.named_parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), k


@pytest.mark.gpu
def test_evaluation_harness_fast_path_equals_the_reference_flow(tmp_path, hip_lib):
    """``evaluate_bayesian_regression_dnn`` (src/evaluation.py:30-108) on the GPU: the fast path (packed parameters,
    device-resident Adam + schedule, one hipGraph replay per step incl. the short last batch, ``DeviceBatches``) against
    the reference's flow (DataLoader, host-side schedule, eager steps) from the same numpy / torch seeds -- same splits,
    same draws in the same order, hence the same test error and test MNLL per split up to float32 reassociation in the
    packed Adam update, and the same checkpoints under the reference's keys."""
    import numpy as np
    from whvi_amd.evaluation import evaluate_bayesian_regression_dnn
    rng = np.random.default_rng(11)
    X = rng.normal(size=(150, 6)).astype(np.float32) * 3 - 1
    y = (np.sin(X[:, :1]) + 0.5 * X[:, 1:2] + 0.05 * rng.normal(size=(150, 1))).astype(np.float32)
    kwargs = dict(epochs1=2, epochs2=6, n_splits=2, batch_size=64, hidden=32, eval_samples=16,
                  optimizer_kwargs={"lambda0": 0.03})             # effective rate 9e-4: the parameters move
    results = {}
    for fast in (False, True):
        np.random.seed(4)
        torch.manual_seed(4)
        results[fast] = evaluate_bayesian_regression_dnn(X, y, "cuda", tmp_path / str(fast), fast=fast, **kwargs)
    assert all(np.isfinite(v) for v in results[True])
    assert np.allclose(results[True], results[False], rtol=2e-4, atol=1e-6), results
    slow = torch.load(tmp_path / "False" / "iter-1" / "epoch-0.pth")          # written after the first epoch of phase two
    quick = torch.load(tmp_path / "True" / "iter-1" / "epoch-0.pth")
    assert list(quick.keys()) == list(slow.keys())
    for k in quick:
        assert torch.allclose(quick[k], slow[k], rtol=1e-4, atol=1e-6), k
