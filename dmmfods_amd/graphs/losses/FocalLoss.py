"""Focal losses on top of per-pixel BCE (reference: dmmfods/graphs/losses/FocalLoss.py:9-91; defined and configured there,
H:125-133, but never instantiated by an agent).  They are ordinary torch modules on the logits returned by the HIP
forward; their backward reaches the HIP backward through the model's autograd bridge (dmm_plan_backward)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _bce(inputs, targets, logits):
    if logits:
        return F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return F.binary_cross_entropy(inputs, targets, reduction="none")


class FocalLoss(nn.Module):
    """F = alpha * (1 - exp(-BCE))**gamma * BCE, unreduced unless ``reduce`` (then the mean)."""

    def __init__(self, alpha=1, gamma=2, logits=False, reduce=True):
        super().__init__()
        self.alpha, self.gamma, self.logits, self.reduce = alpha, gamma, logits, reduce

    def _weights(self, like):
        return self.alpha, self.gamma

    def forward(self, inputs, targets):
        bce = _bce(inputs, targets, self.logits)
        alpha, gamma = self._weights(bce)
        loss = alpha * (1.0 - torch.exp(-bce)) ** gamma * bce
        return loss.mean() if self.reduce else loss


class ClassWiseFocalLoss(FocalLoss):
    """Per-class alpha (class-class imbalance) and gamma (class-background imbalance); inputs are (B, C, H, W)."""

    def __init__(self, alpha=(1, 1, 1), gamma=(2, 2, 2), logits=True, reduce=False):
        super().__init__(list(alpha), list(gamma), logits, reduce)
        if len(self.alpha) != len(self.gamma):
            raise ValueError("alpha and gamma must have the same length")

    def _weights(self, like):
        shape = (1, -1, 1, 1)
        a = torch.as_tensor(self.alpha, dtype=like.dtype, device=like.device).view(shape)
        g = torch.as_tensor(self.gamma, dtype=like.dtype, device=like.device).view(shape)
        return a, g
