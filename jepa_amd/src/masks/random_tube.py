"""Random-tube mask collator (reference src/masks/random_tube.py): alternative collator with the same output
contract as multiblock3d.  Not part of any BASELINE config; host-only integer work."""
from multiprocessing import Value

import numpy as np
import torch


class MaskCollator(object):
    def __init__(self, cfgs_mask, crop_size=(224, 224), num_frames=16, patch_size=(16, 16), tubelet_size=2):
        super(MaskCollator, self).__init__()
        self.mask_generators = [
            _MaskGenerator(crop_size=crop_size, num_frames=num_frames, spatial_patch_size=patch_size,
                           temporal_patch_size=tubelet_size, ratio=m.get('ratio')) for m in cfgs_mask]

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
                 ratio=0.9):
        if not isinstance(crop_size, tuple):
            crop_size = (crop_size,) * 2
        self.height, self.width = crop_size[0] // spatial_patch_size, crop_size[1] // spatial_patch_size
        self.duration = num_frames // temporal_patch_size
        self.num_patches_spatial = self.height * self.width
        self.ratio = ratio
        self.num_keep_spatial = int(self.num_patches_spatial * (1. - self.ratio))
        self._itr_counter = Value('i', -1)

    def step(self):
        with self._itr_counter.get_lock():
            self._itr_counter.value += 1
            return self._itr_counter.value

    def __call__(self, batch_size):
        encs, preds = [], []
        for _ in range(batch_size):
            keep = np.hstack([np.zeros(self.num_patches_spatial - self.num_keep_spatial),
                              np.ones(self.num_keep_spatial)])
            np.random.shuffle(keep)
            keep = torch.tensor(np.tile(keep, (self.duration, 1))).flatten()
            preds.append(torch.nonzero(keep == 0).reshape(-1))
            encs.append(torch.nonzero(keep).reshape(-1))
        return torch.stack(encs), torch.stack(preds)
