"""truncated_exp with the semantics of humanrf/utils/activation.py:6-39 (fp32 exp forward; backward
dy * exp(clamp(x, -15, 15))). The fused kernels implement the same thing in their epilogues
(csrc/mlp.hip); this torch version serves code written against the reference's helper."""
import torch


class _truncated_exp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, threshold):
        x = x.float()
        ctx.save_for_backward(x)
        ctx.threshold = threshold
        return torch.exp(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return dy * torch.exp(x.clamp(-ctx.threshold, ctx.threshold)), None


def truncated_exp(inp: torch.Tensor, threshold: float = 15) -> torch.Tensor:
    return _truncated_exp.apply(inp, threshold)
