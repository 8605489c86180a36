"""Input edge of the step: pinned, double-buffered host-to-device staging of the next batch on a copy stream.

Reference: `load_clips()` (app/vjepa/train.py:391-408) issues `.to(device, non_blocking=True)` for every clip tensor
and every mask right before the step and the step then waits for them: 231 MB of fp32 pixels per step at ViT-L B=24
(PCIe Gen5 x16: >= 3.7 ms) sit on the critical path, 3.7 GB at the ViT-H B=384 recipe.  Here batch k+1 is copied while
step k computes:

    pf = DevicePrefetcher(fetch, device)          # fetch() -> (list of clip tensors [b,3,T,H,W], masks_enc, masks_pred)
    clips, masks_enc, masks_pred = pf.next()      # device tensors of batch k; batch k+1 is already in flight

Two device slots and two pinned host slots per stream of tensors; the copy stream waits until the step that consumed a
slot has been enqueued completely before overwriting it (event recorded on the compute stream at the following
`next()`), and the compute stream waits on the copy's event.  The only host-side wait is on the PINNED staging buffers:
before the host memcpy of batch k+depth into a slot's pinned buffer, the host waits for the event recorded after the
slot's previous H2D copy (batch k) -- normally long complete, so the wait is free, but without it a host that runs more
than `depth` steps ahead of the GPU (loss read every N steps, pageable loader tensors) would overwrite pixels a DMA has
not read yet.  The fp32 -> bf16 cast stays fused in `vj_tubelet_pack` (the copy moves the loader's fp32 pixels verbatim).
"""
import torch

from ..src.utils.tensors import repeat_interleave_batch


class DevicePrefetcher:
    def __init__(self, fetch, device, batch_size=None, num_clips=1, depth=2):
        """fetch(): returns the next host batch `(clip_tensors, masks_enc, masks_pred)` (raises StopIteration when the
        data is exhausted -- the caller's `fetch` normally re-creates its loader instead)."""
        self.fetch, self.device = fetch, torch.device(device)
        self.batch_size, self.num_clips, self.depth = batch_size, num_clips, depth
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._host = [dict() for _ in range(depth)]    # slot -> {key: pinned tensor}
        self._dev = [dict() for _ in range(depth)]     # slot -> {key: device tensor}
        self._free_ev = [None] * depth                 # compute-stream event: slot's previous contents are consumed
        self._h2d_done = [None] * depth                # copy-stream event: the slot's pinned buffers have been read
        self._slot = 0
        self._inflight = None                          # (slot, ready_event, structure)
        self._last_slot = None
        self.bytes_copied = 0

    # -- one tensor through pinned staging into the slot's device buffer
    def _stage(self, slot, key, t):
        t = t.contiguous()
        host, dev = self._host[slot].get(key), self._dev[slot].get(key)
        if dev is None or dev.shape != t.shape or dev.dtype != t.dtype:
            dev = torch.empty(t.shape, dtype=t.dtype, device=self.device)
            self._dev[slot][key] = dev
        if t.is_pinned():
            src = t
        else:
            if host is None or host.shape != t.shape or host.dtype != t.dtype:
                host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                self._host[slot][key] = host
            host.copy_(t)       # pageable -> pinned (host memcpy; loaders with pin_memory=True skip this)
            src = host
        dev.copy_(src, non_blocking=True)
        self.bytes_copied += t.numel() * t.element_size()
        return dev

    def _launch(self):
        clip_list, masks_enc, masks_pred = self.fetch()
        slot = self._slot
        self._slot = (slot + 1) % self.depth
        if self._h2d_done[slot] is not None:
            self._h2d_done[slot].synchronize()         # the DMA engine is done with this slot's pinned staging buffers
        with torch.cuda.stream(self.copy_stream):
            if self._free_ev[slot] is not None:
                self.copy_stream.wait_event(self._free_ev[slot])
            clips = [self._stage(slot, ("clip", i), u) for i, u in enumerate(clip_list)]
            me = [self._stage(slot, ("me", i), m) for i, m in enumerate(masks_enc)]
            mp = [self._stage(slot, ("mp", i), m) for i, m in enumerate(masks_pred)]
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        self._h2d_done[slot] = ready
        self._inflight = (slot, ready, clips, me, mp)

    def next(self, lookahead=True):
        """Device tensors of the next batch (clips concatenated over num_clips, masks batch-repeated like
        train.py:398-406); starts copying the batch after it unless lookahead=False.  The train loop passes
        lookahead=False on the last iteration of an epoch: the reference re-creates its loader iterator at the first
        `next(loader)` of the next epoch, i.e. AFTER `sampler.set_epoch(epoch + 1)` (train.py:366-381) -- fetching across
        the boundary would permute epoch e+1 with epoch e's seed and draw one batch too many at the end of training."""
        cur = torch.cuda.current_stream()
        if self._last_slot is not None:   # everything that read the previous batch has been enqueued by now
            ev = torch.cuda.Event()
            ev.record(cur)
            self._free_ev[self._last_slot] = ev
        if self._inflight is None:
            self._launch()
        slot, ready, clips, me, mp = self._inflight
        self._inflight = None
        cur.wait_event(ready)
        self._last_slot = slot
        if lookahead:
            try:
                self._launch()           # batch k+1 overlaps step k
            except StopIteration:
                self._inflight = None
        clips_d = clips[0] if len(clips) == 1 else torch.cat(clips, dim=0)
        if self.batch_size is not None:
            me = [repeat_interleave_batch(m, self.batch_size, repeat=self.num_clips) for m in me]
            mp = [repeat_interleave_batch(m, self.batch_size, repeat=self.num_clips) for m in mp]
        return clips_d, me, mp
