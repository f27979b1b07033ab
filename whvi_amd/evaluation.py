"""Optimizer set-up of the reference's experiment harness (src/evaluation.py:15-27).

Only ``make_optimizer`` is mirrored: it is part of the training path.  The UCI experiment driver
around it (``evaluate_bayesian_regression_dnn``: sklearn splits, dataset standardisation, eight repetitions) is a
script over this package's public surface and out of scope.
"""
import torch
import torch.optim as optim

__all__ = ["make_optimizer", "DeviceLambdaLR"]


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


def make_optimizer(net, gamma=0.0005, p=0.3, lambda0=0.001, capturable=False, packed=False):
    """Adam plus the decaying schedule of the reference's experiments; returns ``(optimizer, scheduler)``.

    Kept exactly as the reference behaves, including its quirk: ``LambdaLR`` MULTIPLIES the optimizer's base
    learning rate (already ``lambda0``) by the lambda's value, and the reference's lambda contains ``lambda0`` again
    (src/evaluation.py:25-26).  The effective rate at step t is therefore

        lr(t) = lambda0 ** 2 * (1 + gamma * t) ** (-p)        # 1e-6 * ... with the defaults, not 1e-3 * ...

    Training trajectories recorded from the reference (tests/golden/train_golden.npz) are replayed against this.

    ``capturable=True`` (not in the reference) builds the SAME recipe for ``WHVINetwork.train_model(..., graphed=True)``:
    Adam keeps its state and its learning rate on the device and the schedule is a ``DeviceLambdaLR``, so one captured
    hipGraph holds loss, backward, ``optimizer.step()`` and ``scheduler.step()``.  ``packed=True`` first switches every
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
    optimizer = optim.Adam(net.parameters(), lr=torch.tensor(lambda0, dtype=torch.float32, device=device), capturable=True)
    scheduler = DeviceLambdaLR(optimizer, lambda t: lambda0 * torch.pow(1.0 + gamma * t, -p), base_lrs=[lambda0])
    return optimizer, scheduler
