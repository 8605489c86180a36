"""apply_masks with the reference's signature (src/masks/utils.py:11-23), running the bit-exact HIP row gather.

Differentiable: the backward is the row scatter kernel (the reference's gather_backward -> scatter_add_ into
zeros; mask indices are unique per sample by construction of the collator)."""
import torch

from ...hip import ops


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.n = x.shape[1]
        return ops.gather_rows(x.contiguous(), idx.contiguous())

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        return ops.scatter_rows(grad.contiguous(), idx, ctx.n), None


def apply_masks(x, masks, concat=True):
    """x: [B, N, D] GPU tensor; masks: list of int64 [B, K] indices of the tokens to keep."""
    if not x.is_cuda:
        raise ValueError("jepa_amd.apply_masks runs the HIP gather kernel: x must be a GPU tensor (no CPU path)")
    outs = [_GatherRows.apply(x, m) for m in masks]
    return torch.cat(outs, dim=0) if concat else outs
