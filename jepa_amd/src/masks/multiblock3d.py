"""Multi-block 3-D mask collator with the reference's constructor, `.step()` and `__call__` contract
(src/masks/multiblock3d.py:20-203).  Host integer work; it reproduces the reference's index tensors bit for
bit (tests/test_host_parity.py) because it consumes torch's RNG streams in the same order:

  * block SIZE  (t,h,w): three torch.rand(1, generator=g) draws from a generator seeded with a shared step
    counter (identical across ranks and samples of a step);
  * block POSITIONS: per sample and per block, three torch.randint draws (top, left, start) from the GLOBAL
    torch RNG;
  * every sample is truncated to the batch-minimum number of kept / predicted tokens (highest indices dropped).
"""
import math
from logging import getLogger
from multiprocessing import Value

import torch

logger = getLogger()


class MaskCollator(object):
    def __init__(self, cfgs_mask, crop_size=(224, 224), num_frames=16, patch_size=(16, 16), tubelet_size=2):
        super(MaskCollator, self).__init__()
        self.mask_generators = [
            _MaskGenerator(crop_size=crop_size, num_frames=num_frames, spatial_patch_size=patch_size,
                           temporal_patch_size=tubelet_size, spatial_pred_mask_scale=m.get('spatial_scale'),
                           temporal_pred_mask_scale=m.get('temporal_scale'), aspect_ratio=m.get('aspect_ratio'),
                           npred=m.get('num_blocks'), max_context_frames_ratio=m.get('max_temporal_keep', 1.0),
                           max_keep=m.get('max_keep', None))
            for m in cfgs_mask]

    def step(self):
        for g in self.mask_generators:
            g.step()

    def __call__(self, batch):
        collated_batch = torch.utils.data.default_collate(batch)
        masks_enc, masks_pred = [], []
        for g in self.mask_generators:
            e, p = g(len(batch))
            masks_enc.append(e)
            masks_pred.append(p)
        return collated_batch, masks_enc, masks_pred


class _MaskGenerator(object):
    def __init__(self, crop_size=(224, 224), num_frames=16, spatial_patch_size=(16, 16), temporal_patch_size=2,
                 spatial_pred_mask_scale=(0.2, 0.8), temporal_pred_mask_scale=(1.0, 1.0), aspect_ratio=(0.3, 3.0),
                 npred=1, max_context_frames_ratio=1.0, max_keep=None):
        super(_MaskGenerator, self).__init__()
        if not isinstance(crop_size, tuple):
            crop_size = (crop_size,) * 2
        self.crop_size = crop_size
        self.height, self.width = crop_size[0] // spatial_patch_size, crop_size[1] // spatial_patch_size
        self.duration = num_frames // temporal_patch_size
        self.spatial_patch_size, self.temporal_patch_size = spatial_patch_size, temporal_patch_size
        self.aspect_ratio = aspect_ratio
        self.spatial_pred_mask_scale, self.temporal_pred_mask_scale = spatial_pred_mask_scale, temporal_pred_mask_scale
        self.npred = npred
        self.max_context_duration = max(1, int(self.duration * max_context_frames_ratio))
        self.max_keep = max_keep
        self._itr_counter = Value('i', -1)  # shared with loader worker processes, like the reference

    def step(self):
        with self._itr_counter.get_lock():
            self._itr_counter.value += 1
            return self._itr_counter.value

    def _sample_block_size(self, generator, temporal_scale, spatial_scale, aspect_ratio_scale):
        u = torch.rand(1, generator=generator).item()
        t = max(1, int(self.duration * (temporal_scale[0] + u * (temporal_scale[1] - temporal_scale[0]))))
        u = torch.rand(1, generator=generator).item()
        n_keep = int(self.height * self.width * (spatial_scale[0] + u * (spatial_scale[1] - spatial_scale[0])))
        u = torch.rand(1, generator=generator).item()
        ar = aspect_ratio_scale[0] + u * (aspect_ratio_scale[1] - aspect_ratio_scale[0])
        h = min(int(round(math.sqrt(n_keep * ar))), self.height)
        w = min(int(round(math.sqrt(n_keep / ar))), self.width)
        return (t, h, w)

    def _sample_block_mask(self, b_size):
        t, h, w = b_size
        top = torch.randint(0, self.height - h + 1, (1,))
        left = torch.randint(0, self.width - w + 1, (1,))
        start = torch.randint(0, self.duration - t + 1, (1,))
        mask = torch.ones((self.duration, self.height, self.width), dtype=torch.int32)
        mask[start:start + t, top:top + h, left:left + w] = 0
        if self.max_context_duration < self.duration:
            mask[self.max_context_duration:, :, :] = 0
        return mask

    def __call__(self, batch_size):
        g = torch.Generator()
        g.manual_seed(self.step())
        p_size = self._sample_block_size(generator=g, temporal_scale=self.temporal_pred_mask_scale,
                                         spatial_scale=self.spatial_pred_mask_scale,
                                         aspect_ratio_scale=self.aspect_ratio)
        keep_lists, pred_lists = [], []
        n_total = self.duration * self.height * self.width
        min_enc = min_pred = n_total
        while len(keep_lists) < batch_size:
            visible = torch.ones((self.duration, self.height, self.width), dtype=torch.int32)
            for _ in range(self.npred):
                visible *= self._sample_block_mask(p_size)
            visible = visible.flatten()
            kept = torch.nonzero(visible).squeeze()
            if kept.numel() == 0:
                continue  # empty context: redraw this sample
            kept = kept.reshape(-1)
            hidden = torch.nonzero(visible == 0).reshape(-1)
            min_enc, min_pred = min(min_enc, kept.numel()), min(min_pred, hidden.numel())
            keep_lists.append(kept)
            pred_lists.append(hidden)
        if self.max_keep is not None:
            min_enc = min(min_enc, self.max_keep)
        masks_pred = torch.stack([p[:min_pred] for p in pred_lists])
        masks_enc = torch.stack([k[:min_enc] for k in keep_lists])
        return masks_enc, masks_pred
