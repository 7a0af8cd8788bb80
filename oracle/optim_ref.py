"""Oracle: AdamW + HF linear schedule, fp32 CPU.  Test infrastructure only.

Follows the objects the reference's train script builds (multimodal_classifier_train.py:152-164):
torch.optim.AdamW defaults (betas 0.9/0.999, eps 1e-8, weight_decay 0.01) and
transformers.get_scheduler("linear").
"""
import torch


def linear_lr(lr0, t, warmup, total):
    """HF get_linear_schedule_with_warmup lambda; ``warmup`` may be a float (the reference passes 0.15*T)."""
    if t < warmup:
        return lr0 * float(t) / float(max(1, warmup))
    return lr0 * max(0.0, float(total - t) / float(max(1, total - warmup)))


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01):
    """One torch.optim.AdamW update, returns (p, m, v).  ``step`` is 1-based."""
    p = p * (1.0 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
