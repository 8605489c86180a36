"""Pass-through collator with the reference's contract (src/masks/default.py): no masks."""
import torch


class DefaultCollator(object):
    def __call__(self, batch):
        return torch.utils.data.default_collate(batch), None, None
