"""jepa_amd: MI355X-native V-JEPA pretraining step (hand-written gfx950 HIP kernels behind a C ABI).

Package layout mirrors the reference's hot path (app/vjepa, src/models, src/masks, src/utils) so that
`from jepa_amd.src.models.vision_transformer import vit_large` reads like the reference import.
"""
import os as _os

# The step keeps up to five HIP streams busy (dgrad chain, weight gradients / EMA-target forward, gradient communication,
# RCCL's internal stream under torch.distributed, input copies).  ROCm multiplexes streams onto 4 hardware queues by
# default; a fifth stream then shares a queue with another one and the two serialise (measured: the target forward stops
# overlapping the context forward, 88.7 -> 101.2 ms per ViT-L step).  Only effective before the HIP runtime initialises,
# i.e. when jepa_amd is imported before the first GPU call; an explicit user setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.2.0"
