"""bce_loss as humanrf/utils/loss.py:4-10."""
import torch


def bce_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    pred_clamped = torch.clamp(pred, min=0, max=1)
    return -(target * torch.log(pred_clamped + 1e-10) + (1 - target) * torch.log(1 - pred_clamped + 1e-10))
