"""Sin-cos position tables (host, float64 numpy), same values as the reference's
src/models/utils/pos_embs.py:11-99 (checked bit-exact in fp32 by tests/test_host_parity.py)."""
import numpy as np


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    """[sin(pos*w) | cos(pos*w)] with w_i = 10000^(-2i/embed_dim); pos: array of positions -> (M, embed_dim)."""
    if embed_dim % 2 != 0:
        raise AssertionError("embed_dim must be even")
    half = embed_dim // 2
    omega = np.arange(half, dtype=float)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    phase = np.asarray(pos, dtype=float).reshape(-1)[:, None] * omega[None, :]
    return np.concatenate([np.sin(phase), np.cos(phase)], axis=1)


def get_1d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    emb = get_1d_sincos_pos_embed_from_grid(embed_dim, np.arange(grid_size, dtype=float))
    return np.concatenate([np.zeros([1, embed_dim]), emb], axis=0) if cls_token else emb


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    hh, ww = np.meshgrid(np.arange(grid_size, dtype=float), np.arange(grid_size, dtype=float), indexing="ij")
    emb = np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, hh),
                          get_1d_sincos_pos_embed_from_grid(embed_dim // 2, ww)], axis=1)
    return np.concatenate([np.zeros([1, embed_dim]), emb], axis=0) if cls_token else emb


def get_3d_sincos_pos_embed(embed_dim, grid_size, grid_depth, cls_token=False, uniform_power=False):
    """Tokens in (depth, height, width) row-major order; channels [depth | height | width] truncated to
    embed_dim.  uniform_power=True gives each axis ceil(D/6)*2 channels, else D/2, D/4, D/4."""
    dd, hh, ww = np.meshgrid(np.arange(grid_depth, dtype=float), np.arange(grid_size, dtype=float),
                             np.arange(grid_size, dtype=float), indexing="ij")
    if uniform_power:
        h_dim = w_dim = d_dim = int(np.ceil(embed_dim / 6) * 2)
    else:
        h_dim = w_dim = embed_dim // 4
        d_dim = embed_dim // 2
    emb = np.concatenate([get_1d_sincos_pos_embed_from_grid(d_dim, dd), get_1d_sincos_pos_embed_from_grid(h_dim, hh),
                          get_1d_sincos_pos_embed_from_grid(w_dim, ww)], axis=1)[:, :embed_dim]
    return np.concatenate([np.zeros([1, embed_dim]), emb], axis=0) if cls_token else emb
