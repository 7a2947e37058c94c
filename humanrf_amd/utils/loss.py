"""Mask regulariser of the training loss (humanrf/trainer.py:213-215 calls the reference's utils/loss.py helper with the
same name): element-wise binary cross entropy on the clamped accumulated opacity, written out by hand because torch's
own BCE refuses to run under autocast. The fused training step computes it inside k_loss; this is the tensor form for
code written against the reference's helper."""
import torch

_EPS = 1e-10  # keeps both logarithms finite at opacity 0 and 1


def bce_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    opacity = pred.clamp(0.0, 1.0)
    log_hit = torch.log(opacity + _EPS)
    log_miss = torch.log((1.0 - opacity) + _EPS)
    return -(target * log_hit + (1.0 - target) * log_miss)
