"""``MSELoss``: drop-in for ``nn.MSELoss()`` (train.py:498) on the fused HIP loss kernel.

Forward is one streaming pass + a fixed-order reduction; backward re-reads pred/target
and writes ``grad_out * 2 (pred - target) / n`` in one pass (the upstream scalar is read
on the device, so no host sync).  ``sums`` of the last call holds sum|d| and sum d^2, the
two quantities test.py:41-42 accumulates."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class _MSEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, holder):
        out3, _ = ops.mse_fwd_bwd(pred, target, want_grad=False)
        ctx.save_for_backward(pred, target)
        if holder is not None:
            holder.sums = out3
        return out3[0]

    @staticmethod
    def backward(ctx, grad_out):
        pred, target = ctx.saved_tensors
        g = grad_out.contiguous().float()
        _, dpred = ops.mse_fwd_bwd(pred, target, want_grad=True, gscale=1.0, gscale_dev=g, want_sums=False)
        return dpred, None, None


class MSELoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.sums = None  # device tensor [mse, sum|d|, sum d^2] of the last call

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if target.dtype != torch.float32:
            target = target.float()
        return _MSEFunction.apply(pred, target, self)


def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return _MSEFunction.apply(pred, target.float(), None)


def eval_sums(pred: torch.Tensor, target: torch.Tensor):
    """(sum |pred-target|, sum (pred-target)^2, numel) as in test.py:39-43, one kernel pass;
    the two sums are returned as a device tensor so the caller decides when to sync."""
    out3, _ = ops.mse_fwd_bwd(pred, target.float(), want_grad=False)
    return out3[1:], pred.numel()
