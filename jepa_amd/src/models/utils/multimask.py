"""Multi-mask wrappers with the reference's call contract (src/models/utils/multimask.py:11-48).

The reference loops the backbone once per mask; here all masks of a step go through ONE fused chain (rows
concatenated along M) and the per-mask outputs are views of its result -- same values, bigger GEMMs."""
import torch.nn as nn


class MultiMaskWrapper(nn.Module):
    def __init__(self, backbone):
        super().__init__()
        self.backbone = backbone

    def forward(self, x, masks=None):
        if masks is None:
            return self.backbone(x)
        if not isinstance(masks, list):
            masks = [masks]
        return self.backbone.forward_masks(x, masks)


class PredictorMultiMaskWrapper(nn.Module):
    def __init__(self, backbone):
        super().__init__()
        self.backbone = backbone

    def forward(self, ctxt, tgt, masks_ctxt, masks_tgt):
        as_list = lambda v: v if isinstance(v, list) else [v]  # noqa: E731
        return self.backbone.forward_masks(as_list(ctxt), as_list(tgt), as_list(masks_ctxt), as_list(masks_tgt))
