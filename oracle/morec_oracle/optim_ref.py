"""AdamW restatement -- TEST INFRASTRUCTURE ONLY.

Follows the published ``torch.optim.AdamW`` single-tensor algorithm that the reference calls at
``T/run.py:159-162,246`` (betas (0.9, 0.999), eps 1e-8, decoupled weight decay, bias correction
on both moments, ``denom = sqrt(v)/sqrt(1-b2^t) + eps``)."""
from __future__ import annotations

import math

import torch


def adamw_step(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
               step: int, lr: float, weight_decay: float, beta1: float = 0.9, beta2: float = 0.999,
               eps: float = 1e-8):
    """One in-place AdamW update; ``step`` is the 1-based step count AFTER increment."""
    param.mul_(1.0 - lr * weight_decay)
    exp_avg.mul_(beta1).add_(grad, alpha=1.0 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-(lr / bc1))
    return param
