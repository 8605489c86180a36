"""jepa_amd: MI355X-native V-JEPA pretraining step (hand-written gfx950 HIP kernels behind a C ABI).

Package layout mirrors the reference's hot path (app/vjepa, src/models, src/masks, src/utils) so that
`from jepa_amd.src.models.vision_transformer import vit_large` reads like the reference import.
"""
__version__ = "0.1.0"
