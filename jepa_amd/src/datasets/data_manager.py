"""Data-loader factory with the reference's `init_data` signature (src/datasets/data_manager.py:15-91).

The reference's CPU video pipeline (decord decode, clip sampling, augmentation: src/datasets/video_dataset.py,
app/vjepa/transforms.py) sits OUTSIDE the accelerated hot path and needs packages that are not part of this
image; `data='synthetic'` provides the seeded synthetic clip stream used by BASELINE.json's configs and the
tests.  Any real dataset type raises with an explanation instead of silently degrading.
"""
import torch


class SyntheticClips(torch.utils.data.Dataset):
    """Seeded N(0,1) clips in the item layout of the reference's VideoDataset (video_dataset.py:156-184):
    ([clip[3,T,H,W]] * num_clips, label, [frame indices])."""

    def __init__(self, length, num_frames, crop_size, num_clips=1, seed=1234):
        self.length, self.num_frames, self.crop_size, self.num_clips, self.seed = length, num_frames, crop_size, num_clips, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        clips = [torch.randn(3, self.num_frames, self.crop_size, self.crop_size, generator=g)
                 for _ in range(self.num_clips)]
        return clips, 0, [torch.arange(self.num_frames) for _ in range(self.num_clips)]


def init_data(batch_size, transform=None, shared_transform=None, data='ImageNet', collator=None, pin_mem=True,
              num_workers=8, world_size=1, rank=0, root_path=None, image_folder=None, training=True, copy_data=False,
              drop_last=True, tokenize_txt=True, subset_file=None, clip_len=8, frame_sample_rate=2, duration=None,
              num_clips=1, random_clip_sampling=True, allow_clip_overlap=False, filter_short_videos=False,
              filter_long_videos=int(1e9), decode_one_clip=True, datasets_weights=None, persistent_workers=False,
              repeat_wds=False, ipe=300, log_dir=None, crop_size=224, synthetic_length=None):
    if str(data).lower() != 'synthetic':
        raise NotImplementedError(
            f"dataset_type={data!r}: the reference's decord/torchvision CPU video pipeline is outside the "
            "accelerated V-JEPA step and its dependencies are not available here; use dataset_type: synthetic, "
            "or assign your own loader factory to jepa_amd.app.vjepa.train.init_data")
    length = synthetic_length if synthetic_length is not None else batch_size * world_size * ipe
    dataset = SyntheticClips(length, clip_len, crop_size, num_clips=num_clips)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world_size, rank=rank, shuffle=True)
    loader = torch.utils.data.DataLoader(dataset, collate_fn=collator, sampler=sampler, batch_size=batch_size,
                                         drop_last=drop_last, pin_memory=pin_mem, num_workers=num_workers,
                                         persistent_workers=(num_workers > 0) and persistent_workers)
    return loader, sampler
