"""Activation modules of the reference (src/activations.py): ``Cosine``, the random-feature non-linearity of the
paper's kernel-approximation experiments.  Elementwise, so it broadcasts over the leading Monte-Carlo sample axis of the
batched passes (``WHVINetwork.forward_batched``) like ``nn.ReLU``."""
import torch
import torch.nn as nn

__all__ = ["Cosine"]


class Cosine(nn.Module):
    """``forward(x) = cos(x)`` (src/activations.py:5-13)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.cos(x)
