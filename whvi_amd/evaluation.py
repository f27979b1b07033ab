"""The reference's experiment harness (src/evaluation.py): ``make_optimizer`` (:15-27, part of the training path, SURVEY.md
F4) and ``evaluate_bayesian_regression_dnn`` (:30-108), the caller of ``train_model`` / ``eval_model`` that the UCI
regression scripts run (experiments/regression_experiments/run_*.py) -- the same protocol, with the training loop on the
fast path (packed parameters, device-resident Adam + schedule, one hipGraph replay per step, batches sliced on the
device) when the device is a GPU.  The data sets themselves are not part of this repo (no network; the scripts download
them or read ``../datasets``).
"""
import pathlib

import torch
import torch.nn as nn
import torch.optim as optim

__all__ = ["make_optimizer", "DeviceLambdaLR", "DeviceDecayLR", "DeviceBatches", "evaluate_bayesian_regression_dnn"]


class DeviceLambdaLR:
    """``torch.optim.lr_scheduler.LambdaLR`` with its state ON THE DEVICE: the step counter and every group's learning
    rate are device tensors and ``step()`` is a handful of tiny device ops into which no host value enters.

    Why: the reference's recipe calls ``scheduler.step()`` after EVERY batch (src/networks.py:80-81,
    src/evaluation.py:25-26).  A host-side ``LambdaLR`` writes a new Python float into ``param_group["lr"]``, which a
    captured hipGraph cannot see -- ``GraphedTrainStep`` could therefore not run the recipe the reference runs.  With
    this class (and ``Adam(..., lr=<tensor>, capturable=True)``, which reads the rate from device memory) the schedule is
    part of the captured step and advances itself on every replay.

    ``factor(t)`` receives the step counter as a 0-d float64 device tensor and returns the multiplier of the group's
    base rate, like ``LambdaLR``'s lambda; as with ``LambdaLR`` the constructor applies ``factor(0)``.
    """
    device_resident = True

    def __init__(self, optimizer, factor, base_lrs=None):
        """``base_lrs``: the groups' base rates as Python floats (default: read from the optimizer -- pass them when the
        optimizer already holds float32 tensors, whose values are the rounded rates)."""
        self.optimizer, self.factor = optimizer, factor
        params = [p for g in optimizer.param_groups for p in g["params"]]
        device = params[0].device
        self.t = torch.zeros((), dtype=torch.float64, device=device)
        self.base_lrs = []
        for i, group in enumerate(optimizer.param_groups):
            self.base_lrs.append(float(group["lr"]) if base_lrs is None else float(base_lrs[i]))
            if not torch.is_tensor(group["lr"]) or group["lr"].device != device:
                group["lr"] = torch.tensor(float(group["lr"]), dtype=torch.float32, device=device)
        self._apply()

    def _apply(self):
        f = self.factor(self.t)
        for group, base in zip(self.optimizer.param_groups, self.base_lrs):
            group["lr"].copy_(base * f)                      # float64 -> the rate's float32, on the device

    def step(self):
        self.t.add_(1.0)
        self._apply()

    def get_last_lr(self):
        """Python floats (synchronises: monitoring and tests only)."""
        return [float(group["lr"]) for group in self.optimizer.param_groups]

    def state_dict(self):
        return {"t": float(self.t), "base_lrs": list(self.base_lrs)}

    def load_state_dict(self, state):
        self.t.fill_(float(state["t"]))
        self.base_lrs = list(state["base_lrs"])
        self._apply()


class DeviceDecayLR(DeviceLambdaLR):
    """``DeviceLambdaLR`` for the one schedule the reference's experiments use -- ``lambda t: lambda0 * (1 + gamma t)^-p``
    (src/evaluation.py:25-26) -- with ``step()`` as ONE single-thread launch per parameter group
    (``whvi_decay_lr_step``) instead of seven tiny float64 tensor ops: in a captured training step of ~60 launches every
    graph node is 2 us."""

    def __init__(self, optimizer, lambda0, gamma, p, base_lrs=None):
        self.lambda0, self.gamma, self.p = float(lambda0), float(gamma), float(p)
        super().__init__(optimizer, lambda t: self.lambda0 * torch.pow(1.0 + self.gamma * t, -self.p), base_lrs=base_lrs)

    def _launch(self, advance):
        from whvi_amd import _hip
        for i, (group, base) in enumerate(zip(self.optimizer.param_groups, self.base_lrs)):
            _hip.decay_lr_step(self.t, group["lr"], base, self.lambda0, self.gamma, self.p, advance=advance and i == 0)

    def _apply(self):
        self._launch(False)

    def step(self):
        self._launch(True)


def make_optimizer(net, gamma=0.0005, p=0.3, lambda0=0.001, capturable=False, packed=False, fused=None):
    """Adam plus the decaying schedule of the reference's experiments; returns ``(optimizer, scheduler)``.

    Kept exactly as the reference behaves, including its quirk: ``LambdaLR`` MULTIPLIES the optimizer's base
    learning rate (already ``lambda0``) by the lambda's value, and the reference's lambda contains ``lambda0`` again
    (src/evaluation.py:25-26).  The effective rate at step t is therefore

        lr(t) = lambda0 ** 2 * (1 + gamma * t) ** (-p)        # 1e-6 * ... with the defaults, not 1e-3 * ...

    Training trajectories recorded from the reference (tests/golden/train_golden.npz) are replayed against this.

    ``capturable=True`` (not in the reference) builds the SAME recipe for ``WHVINetwork.train_model(..., graphed=True)``:
    Adam (torch's fused multi-tensor kernel unless ``fused=False``) keeps its state and its learning rate on the device and
    the schedule is a ``DeviceDecayLR`` (one launch per step), so one captured hipGraph holds loss, backward,
    ``optimizer.step()`` and ``scheduler.step()``.  ``packed=True`` first switches every
    stacked layer to the packed parameter layout (``WHVINetwork.pack_parameters``: checkpoints keep the reference's keys).

    :param net: target model.
    :param gamma: decay parameter.
    :param p: decay parameter.
    :param lambda0: learning rate (enters the effective rate squared, see above).
    """
    if packed:
        net.pack_parameters()
    if not capturable:
        optimizer = optim.Adam(net.parameters(), lr=lambda0)
        scheduler = optim.lr_scheduler.LambdaLR(optimizer, lambda t: lambda0 * ((1 + gamma * t) ** (-p)))
        return optimizer, scheduler
    device = next(net.parameters()).device
    if device.type != "cuda":
        raise RuntimeError("make_optimizer(capturable=True) needs the network on a GPU")
    # fused=True: torch's one-launch multi-tensor Adam.  The default (foreach) implementation with capturable=True spends
    # ~42 launches per step on this network -- among them one broadcast division PER PARAMETER TENSOR for each of the two
    # bias corrections, whose 0-d step tensors take _foreach_div_ off its fast path -- i.e. 40 % of a captured step
    # (profiles/r03/train_graph_before_after.log).  Same update formula; float32 rounding of intermediate terms may differ.
    fused = True if fused is None else bool(fused)
    optimizer = optim.Adam(net.parameters(), lr=torch.tensor(lambda0, dtype=torch.float32, device=device), capturable=True,
                           fused=fused)
    scheduler = DeviceDecayLR(optimizer, lambda0, gamma, p, base_lrs=[lambda0])
    return optimizer, scheduler


class DeviceBatches:
    """The batches ``DataLoader(TensorDataset(X, y), batch_size=b)`` yields -- in order, the last one short, exactly what
    the reference's harness trains on (src/evaluation.py:73-74: no shuffling) -- as VIEWS of the two tensors.

    Why: with the step itself a 0.3 ms hipGraph replay, ``DataLoader`` is the bottleneck by an order of magnitude (it
    indexes the data set row by row and collates 64 one-row tensors per batch, on the device: ~130 tiny launches).  The
    views are made once; ``train_model`` only needs iteration and ``len(loader.dataset)``."""

    def __init__(self, X: torch.Tensor, y: torch.Tensor, batch_size: int = 64):
        if X.size(0) != y.size(0):
            raise ValueError("DeviceBatches: X and y differ in their number of rows")
        from torch.utils.data import TensorDataset
        self.dataset = TensorDataset(X, y)
        self.batch_size = int(batch_size)
        self._batches = [(X[i:i + self.batch_size], y[i:i + self.batch_size]) for i in range(0, X.size(0), self.batch_size)]

    def __iter__(self):
        # DataLoader's iterator draws its base seed from the host generator once per pass (torch/utils/data/dataloader.py,
        # _BaseDataLoaderIter.__init__); the same draw here, so that whatever runs after the loop -- the next split's
        # parameter initialisation in the harness below -- sees the same generator state under either loader
        torch.empty((), dtype=torch.int64).random_()
        return iter(self._batches)

    def __len__(self):
        return len(self._batches)


def evaluate_bayesian_regression_dnn(X, y, device, checkpoint_dir, *, epochs1: int = 500, epochs2: int = 50000,
                                     n_splits: int = 8, batch_size: int = 64, hidden: int = 128, eval_samples: int = 64,
                                     fast=None, pbar_update_period=None, random_state=None, optimizer_kwargs=None):
    """Test error (RMSE of the predictive mean) and test MNLL on the data set ``(X, y)``: the protocol of
    src/evaluation.py:30-108 (section 3.2 of the paper, D.1 of its supplement), returned as ``(error_mean, error_sd,
    mnll_mean, mnll_sd)`` over the random splits.

    As in the reference: network ``(n_in, 128, 128, n_out)`` of ``WHVILinear`` layers with ``lambda_=3.0`` on the two
    hidden ones and ReLU between, 64 MC samples at test time and 1 in training; columns of ``X`` standardised here
    (over the WHOLE data set, before splitting -- as written in the reference), targets unchanged; eight random 90 % /
    10 % splits (``sklearn.model_selection.train_test_split`` on numpy's global generator unless ``random_state`` is
    given); ``make_optimizer`` defaults (Adam, lambda0 = 0.001 -- entering squared, p = 0.3, gamma = 0.0005); batches of
    64 in data order; 500 epochs with the two-phase labels of ``train_model`` and 50 000 more, a checkpoint of the second
    phase every 5 000 epochs under ``checkpoint_dir/iter-{k}/epoch-{e}.pth``.

    The keyword arguments are not in the reference (its values are the defaults): ``epochs1`` / ``epochs2`` /
    ``n_splits`` / ``batch_size`` / ``hidden`` / ``eval_samples`` for smaller runs; ``fast`` -- ``None`` = on a GPU --
    selects ``make_optimizer(capturable=True, packed=True)``, ``train_model(graphed=True)`` and ``DeviceBatches``: the same
    update per step WITHIN A TOLERANCE, not bit for bit -- the fast path keeps the learning rate as a float32 device scalar
    evaluated with the device's pow() (``DeviceDecayLR``: float64 arithmetic, ONE rounding to float32, <= 1.2e-7 relative
    to the reference's Python-float LambdaLR rate over 3 000 steps, tests/test_config_parity.py::
    test_one_launch_schedule_is_the_reference_schedule_gpu) and torch's fused Adam may round intermediate terms
    differently; the parity test of the two flows covers a handful of steps
    (tests/test_config_parity.py::test_evaluation_harness_fast_path_equals_the_reference_flow), not the 252 500-step
    protocol, over which such differences compound like any float32 training noise.  ``fast=False`` is the reference's flow.  The
    308-row yacht data set's 252 500 steps per split in minutes instead of an hour; ``pbar_update_period`` (reference: 1,
    i.e. one device-to-host read of KL and MNLL per epoch; fast path default 500); ``optimizer_kwargs`` for
    ``make_optimizer`` (``lambda0`` enters the rate squared -- the defaults give 1e-6)."""
    import numpy as np
    from sklearn.model_selection import train_test_split
    from sklearn.preprocessing import StandardScaler
    from torch.utils.data import DataLoader, TensorDataset
    from whvi_amd.layers import WHVILinear
    from whvi_amd.networks import WHVIRegression

    assert len(y.shape) == 2
    assert len(X.shape) == 2
    assert len(X) == len(y)
    device = torch.device(device)
    if fast is None:
        fast = device.type == "cuda"
    if fast and device.type != "cuda":
        raise RuntimeError("evaluate_bayesian_regression_dnn(fast=True) needs a GPU device")
    if pbar_update_period is None:
        pbar_update_period = 500 if fast else 1

    test_errors, test_mnlls = [], []
    X = StandardScaler().fit_transform(X)
    for index in range(n_splits):
        print(f'Iteration {index + 1}/{n_splits}')
        X_train, X_test, y_train, y_test = train_test_split(X, y, train_size=0.9, test_size=0.1, random_state=random_state)
        X_train, y_train = torch.tensor(X_train, device=device), torch.tensor(y_train, device=device)
        X_test, y_test = torch.tensor(X_test, device=device), torch.tensor(y_test, device=device)
        if fast:
            train_loader = DeviceBatches(X_train, y_train, batch_size=batch_size)
        else:
            train_loader = DataLoader(TensorDataset(X_train, y_train), batch_size=batch_size)

        model = WHVIRegression([
            WHVILinear(X_test.size()[1], hidden, lambda_=3.0),
            nn.ReLU(),
            WHVILinear(hidden, hidden, lambda_=3.0),
            nn.ReLU(),
            WHVILinear(hidden, y_test.size()[1])
        ], eval_samples=eval_samples)
        model = model.to(device=device)
        optimizer, scheduler = make_optimizer(model, capturable=bool(fast), packed=bool(fast), **(optimizer_kwargs or {}))

        iteration_dir = pathlib.Path(checkpoint_dir) / f'iter-{index}'
        iteration_dir.mkdir(exist_ok=True, parents=True)
        model.train_model(train_loader, optimizer, scheduler, epochs1=epochs1, epochs2=epochs2,
                          pbar_update_period=pbar_update_period, checkpoint_dir=iteration_dir, graphed=bool(fast),
                          sharded=False)

        with torch.no_grad():
            error, mnll = model.eval_model(X_test, y_test)
        print(f"Error: {error}, MNLL: {mnll}")
        test_errors.append(error)
        test_mnlls.append(mnll)

    return float(np.mean(test_errors)), float(np.std(test_errors)), float(np.mean(test_mnlls)), float(np.std(test_mnlls))
