"""Optimizer set-up of the reference's experiment harness (src/evaluation.py:15-27).

Only ``make_optimizer`` is mirrored: it is part of the training path.  The UCI experiment driver
around it (``evaluate_bayesian_regression_dnn``: sklearn splits, dataset standardisation, eight repetitions) is a
script over this package's public surface and out of scope.
"""
import torch.optim as optim

__all__ = ["make_optimizer"]


def make_optimizer(net, gamma=0.0005, p=0.3, lambda0=0.001):
    """Adam plus the decaying schedule of the reference's experiments; returns ``(optimizer, scheduler)``.

    Kept exactly as the reference behaves, including its quirk: ``LambdaLR`` MULTIPLIES the optimizer's base
    learning rate (already ``lambda0``) by the lambda's value, and the reference's lambda contains ``lambda0`` again
    (src/evaluation.py:25-26).  The effective rate at step t is therefore

        lr(t) = lambda0 ** 2 * (1 + gamma * t) ** (-p)        # 1e-6 * ... with the defaults, not 1e-3 * ...

    Training trajectories recorded from the reference (tests/golden/train_golden.npz) are replayed against this.

    :param net: target model.
    :param gamma: decay parameter.
    :param p: decay parameter.
    :param lambda0: learning rate (enters the effective rate squared, see above).
    """
    optimizer = optim.Adam(net.parameters(), lr=lambda0)
    scheduler = optim.lr_scheduler.LambdaLR(optimizer, lambda t: lambda0 * ((1 + gamma * t) ** (-p)))
    return optimizer, scheduler
