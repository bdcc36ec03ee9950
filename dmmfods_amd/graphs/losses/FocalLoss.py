"""Focal losses on top of per-pixel BCE (reference: dmmfods/graphs/losses/FocalLoss.py:9-91; defined and configured there,
H:125-133, but never instantiated by an agent).  The arithmetic is the loss epilogue of the HIP BCE/metrics kernel
(``bce_metrics_kernel``, kind DMM_LOSS_FOCAL):

  * ``model.set_loss("focal", alpha, gamma)`` (or ``loss.attach(model)``) makes the fused training tail
    ``model.loss_backward`` / ``model.loss_metrics`` compute the focal loss sums and backpropagate d(sum F)/d(logit);
  * calling the module on device tensors runs the same kernel stand-alone (``dmm_loss_forward``): unreduced loss forward,
    the analytic derivative the kernel produced in backward.  There is no CPU path.  The kernel computes in fp32 whatever the
    input dtype (fp64 inputs are rounded to fp32 first and the result is cast back: the reference's fp64 precision is NOT kept).
"""
import ctypes as C

import torch
import torch.nn as nn

from ... import _lib


class _HipLoss(torch.autograd.Function):
    """Unreduced loss of `dmm_loss_forward` on a (B, C, H, W) view; alpha / gamma per class (channels are processed eight at a time,
    the kernel's limit).  Returns the input's dtype, like the torch expression of the reference."""

    @staticmethod
    def forward(ctx, inputs, targets, kind, from_prob, alpha, gamma):
        if not inputs.is_cuda:
            raise RuntimeError("dmmfods_amd computes on the GPU only; move the tensors to 'cuda' (no CPU fallback)")
        if inputs.shape != targets.shape:
            raise ValueError("expected inputs and targets of the same shape")
        B, NC, H, W = inputs.shape
        x = inputs.detach().contiguous().float()
        t = targets.detach().contiguous().float()
        loss = torch.empty_like(x)
        dx = torch.empty_like(x)
        L = _lib.lib()
        if NC <= 8:
            a = (C.c_float * NC)(*alpha)
            g = (C.c_float * NC)(*gamma)
            _lib.check(L.dmm_loss_forward(kind, 1 if from_prob else 0, a, g, x.data_ptr(), t.data_ptr(), loss.data_ptr(),
                                          dx.data_ptr(), B, NC, H, W, _lib.stream_ptr()))
        else:   # eight channels per launch on contiguous copies of the channel groups
            for c0 in range(0, NC, 8):
                n = min(8, NC - c0)
                xs, ts = x[:, c0:c0 + n].contiguous(), t[:, c0:c0 + n].contiguous()
                ls, ds = torch.empty_like(xs), torch.empty_like(xs)
                a = (C.c_float * n)(*alpha[c0:c0 + n])
                g = (C.c_float * n)(*gamma[c0:c0 + n])
                _lib.check(L.dmm_loss_forward(kind, 1 if from_prob else 0, a, g, xs.data_ptr(), ts.data_ptr(), ls.data_ptr(),
                                              ds.data_ptr(), B, n, H, W, _lib.stream_ptr()))
                loss[:, c0:c0 + n], dx[:, c0:c0 + n] = ls, ds
        ctx.save_for_backward(dx)
        ctx.in_dtype = inputs.dtype
        return loss.to(inputs.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        (dx,) = ctx.saved_tensors
        return (grad_out.float() * dx).to(ctx.in_dtype), None, None, None, None, None


class FocalLoss(nn.Module):
    """F = alpha * (1 - exp(-BCE))**gamma * BCE, unreduced unless ``reduce`` (then the mean), on tensors of ANY shape (reference
    L:30-50).  ``logits=False``: the inputs are probabilities (F.binary_cross_entropy), as in the reference's default."""

    def __init__(self, alpha=1, gamma=2, logits=False, reduce=True):
        super().__init__()
        self.alpha, self.gamma, self.logits, self.reduce = alpha, gamma, logits, reduce

    @staticmethod
    def _per_class(nclass, alpha, gamma, exact=True):
        """alpha / gamma as lists of `nclass` floats (a scalar is repeated; a shorter list is padded with the values that give zero
        loss when ``exact`` is False).  Pure function of its arguments: forward() never rebinds the module's attributes, so concurrent
        calls on one module cannot see each other's values."""
        def expand(v, pad):
            v = [float(v)] * nclass if not hasattr(v, "__len__") else [float(e) for e in v]
            if len(v) > nclass or (exact and len(v) != nclass):
                raise ValueError(f"expected {nclass} per-class values, got {len(v)}")
            return v + [pad] * (nclass - len(v))
        return expand(alpha, 0.0), expand(gamma, 1.0)

    def attach(self, model):
        """Select this loss as the epilogue of the model's fused training tail (logits only, as the tail sees logits)."""
        if not self.logits:
            raise ValueError("the fused training tail works on logits: construct the loss with logits=True")
        alpha, gamma = self._per_class(int(model.num_classes), self.alpha, self.gamma)
        model.set_loss("focal", alpha, gamma)
        return model

    @staticmethod
    def _is_scalar(v):
        return isinstance(v, (int, float)) or (torch.is_tensor(v) and v.dim() == 0)

    def forward(self, inputs, targets):
        shape = inputs.shape
        x = inputs.reshape(1, 1, 1, -1)
        t = targets.reshape(1, 1, 1, -1)
        if self._is_scalar(self.alpha) and self._is_scalar(self.gamma):
            # scalar alpha / gamma act element-wise: any shape is one "class" of numel elements, all of it in the HIP epilogue
            loss = _HipLoss.apply(x, t, _lib.LOSS_FOCAL, not self.logits, [float(self.alpha)], [float(self.gamma)]).reshape(shape)
        else:
            # tensor / sequence alpha or gamma: the reference's expression (L:44-45) broadcasts them against the loss tensor.  The
            # unreduced BCE (and its derivative) come from the HIP kernel; the broadcast itself is torch glue on the device.
            bce = _HipLoss.apply(x, t, _lib.LOSS_BCE, not self.logits, [1.0], [0.0]).reshape(shape)
            alpha = torch.as_tensor(self.alpha, dtype=bce.dtype, device=bce.device)
            gamma = torch.as_tensor(self.gamma, dtype=bce.dtype, device=bce.device)
            loss = alpha * (1 - torch.exp(-bce)) ** gamma * bce
        return loss.mean() if self.reduce else loss


class ClassWiseFocalLoss(FocalLoss):
    """Per-class alpha (class-class imbalance) and gamma (class-background imbalance); inputs are (B, C, H, W).  As in the reference
    (L:78-91, a loop over zip(alpha, gamma)): lists of different lengths are cut to the shorter one, classes beyond the listed
    values get zero loss, more listed values than channels raise IndexError."""

    def __init__(self, alpha=(1, 1, 1), gamma=(2, 2, 2), logits=True, reduce=False):
        super().__init__(list(alpha), list(gamma), logits, reduce)

    def forward(self, inputs, targets):
        if inputs.dim() != 4:
            raise ValueError("expected inputs and targets structured like batches x classes x X x Y")
        n = min(len(self.alpha), len(self.gamma))   # zip() stops at the shorter list
        if n > inputs.shape[1]:
            raise IndexError(f"{n} per-class values for {inputs.shape[1]} classes")   # the reference's F_loss[:, i] raises the same
        alpha, gamma = self._per_class(inputs.shape[1], list(self.alpha)[:n], list(self.gamma)[:n], exact=False)
        loss = _HipLoss.apply(inputs, targets, _lib.LOSS_FOCAL, not self.logits, alpha, gamma)
        return loss.mean() if self.reduce else loss

    def attach(self, model):
        if not self.logits:
            raise ValueError("the fused training tail works on logits: construct the loss with logits=True")
        n = min(len(self.alpha), len(self.gamma))
        alpha, gamma = self._per_class(int(model.num_classes), list(self.alpha)[:n], list(self.gamma)[:n], exact=False)
        model.set_loss("focal", alpha, gamma)
        return model
