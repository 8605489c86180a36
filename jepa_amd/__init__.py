"""jepa_amd: MI355X-native V-JEPA pretraining step (hand-written gfx950 HIP kernels behind a C ABI).

Package layout mirrors the reference's hot path (app/vjepa, src/models, src/masks, src/utils) so that
`from jepa_amd.src.models.vision_transformer import vit_large` reads like the reference import.
"""
import os as _os

# Hardware queues.  ROCclr multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (ROCm default 4); two streams on one
# queue serialise.  The step keeps four streams of its own busy (dgrad chain, weight gradients / EMA-target forward, the deferred
# update, gradient communication) -- each is CHOSEN so that it shares a queue with none of the others (engine/layers.py
# independent_stream) -- next to the streams torch.distributed and RCCL create for themselves.  Round 2 set 8 (a fifth stream on four
# queues serialised the target forward, before streams were checked); round 5 measured the whole matrix at one RCCL rank
# (profiles/r05_dp1_coll_mode.md): 2, 4 and 6 queues run the plain step, the torch.distributed reducer and the C-ABI reducer at the
# same 70.4 - 70.5 ms; with 8 (and 24) a SECOND RCCL communicator in the process -- the C-ABI route's -- costs every compute kernel
# 20 - 50 % (86.2 ms), whether or not it is ever used.  6 leaves room for an input-copy stream and stays clear of that state.
# Only effective before the HIP runtime initialises, i.e. when jepa_amd is imported before the first GPU call; an explicit user
# setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")

__version__ = "0.2.0"
