"""Focal losses on top of per-pixel BCE (reference: dmmfods/graphs/losses/FocalLoss.py:9-91; defined and configured there,
H:125-133, but never instantiated by an agent).  The arithmetic is the loss epilogue of the HIP BCE/metrics kernel
(``bce_metrics_kernel``, kind DMM_LOSS_FOCAL):

  * ``model.set_loss("focal", alpha, gamma)`` (or ``loss.attach(model)``) makes the fused training tail
    ``model.loss_backward`` / ``model.loss_metrics`` compute the focal loss sums and backpropagate d(sum F)/d(logit);
  * calling the module on device tensors runs the same kernel stand-alone (``dmm_loss_forward``): unreduced loss forward,
    the analytic derivative the kernel produced in backward.  There is no CPU path.
"""
import ctypes as C

import torch
import torch.nn as nn

from ... import _lib


class _HipLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, kind, from_prob, alpha, gamma):
        if not inputs.is_cuda:
            raise RuntimeError("dmmfods_amd computes on the GPU only; move the tensors to 'cuda' (no CPU fallback)")
        if inputs.dim() != 4 or inputs.shape != targets.shape:
            raise ValueError("expected inputs and targets of the same (batches, classes, X, Y) shape")
        B, NC, H, W = inputs.shape
        if NC > 8:
            raise ValueError("at most 8 classes are supported")
        x = inputs.detach().contiguous().float()
        t = targets.detach().contiguous().float()
        loss = torch.empty_like(x)
        dx = torch.empty_like(x)
        a = (C.c_float * NC)(*alpha)
        g = (C.c_float * NC)(*gamma)
        _lib.check(_lib.lib().dmm_loss_forward(kind, 1 if from_prob else 0, a, g, x.data_ptr(), t.data_ptr(), loss.data_ptr(),
                                               dx.data_ptr(), B, NC, H, W, _lib.stream_ptr()))
        ctx.save_for_backward(dx)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (dx,) = ctx.saved_tensors
        return grad_out * dx, None, None, None, None, None


class FocalLoss(nn.Module):
    """F = alpha * (1 - exp(-BCE))**gamma * BCE, unreduced unless ``reduce`` (then the mean).  ``logits=False``: the inputs
    are probabilities (F.binary_cross_entropy), as in the reference's default."""

    def __init__(self, alpha=1, gamma=2, logits=False, reduce=True):
        super().__init__()
        self.alpha, self.gamma, self.logits, self.reduce = alpha, gamma, logits, reduce

    def _per_class(self, nclass):
        def expand(v):
            v = [float(v)] * nclass if not hasattr(v, "__len__") else [float(e) for e in v]
            if len(v) != nclass:
                raise ValueError(f"expected {nclass} per-class values, got {len(v)}")
            return v
        return expand(self.alpha), expand(self.gamma)

    def attach(self, model):
        """Select this loss as the epilogue of the model's fused training tail (logits only, as the tail sees logits)."""
        if not self.logits:
            raise ValueError("the fused training tail works on logits: construct the loss with logits=True")
        alpha, gamma = self._per_class(int(model.num_classes))
        model.set_loss("focal", alpha, gamma)
        return model

    def forward(self, inputs, targets):
        alpha, gamma = self._per_class(inputs.shape[1] if inputs.dim() == 4 else 0)
        loss = _HipLoss.apply(inputs, targets, _lib.LOSS_FOCAL, not self.logits, alpha, gamma)
        return loss.mean() if self.reduce else loss


class ClassWiseFocalLoss(FocalLoss):
    """Per-class alpha (class-class imbalance) and gamma (class-background imbalance); inputs are (B, C, H, W)."""

    def __init__(self, alpha=(1, 1, 1), gamma=(2, 2, 2), logits=True, reduce=False):
        super().__init__(list(alpha), list(gamma), logits, reduce)
        if len(self.alpha) != len(self.gamma):
            raise ValueError("alpha and gamma must have the same length")
