"""The input edge (engine/input.py DevicePrefetcher; reference app/vjepa/train.py:391-408) and the stand-alone module forwards / frozen-encoder
inference (src/models/utils/modules.py, vision_transformer.py) against fp32 PyTorch and the oracle."""
import os
import socket
import sys
import pytest
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import HP, rel_l2  # noqa: E402
from tests.step_util import (TINY, TINY_MASKS, build_models, build_trainer, draw_batch, oracle_cfg,  # noqa: E402
                             to_dev)
from tests.golden_util import rel_l2  # noqa: E402
from tests.step_util import TINY, TINY_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402
import math

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


DEV = "cuda"


# ------------------------------------------------------------------------------------------------ input edge
def test_device_prefetcher_delivers_every_batch_in_order():
    """Pinned double-buffered H2D on a copy stream: contents bit-exact, order preserved, slots safely reused while a
    consumer kernel is still running on the compute stream."""
    from jepa_amd.engine.input import DevicePrefetcher
    host = []
    g = torch.Generator().manual_seed(3)
    for i in range(7):
        host.append(([torch.randn(2, 3, 4, 16, 16, generator=g)], [torch.randint(0, 50, (2, 5 + i))],
                     [torch.randint(0, 50, (2, 9))]))
    it = iter(host)
    pf = DevicePrefetcher(lambda: next(it), torch.device(DEV), batch_size=2, num_clips=1)
    busy = torch.randn(4096, 4096, device=DEV)
    for i in range(7):
        clips, me, mp = pf.next()
        acc = clips.clone()
        for _ in range(3):
            busy = busy @ busy * 1e-3            # keep the compute stream busy while the next copy is in flight
        acc2 = clips.clone()                     # read again AFTER the busy work: the slot must not have been overwritten
        torch.cuda.synchronize()
        assert torch.equal(acc.cpu(), host[i][0][0]) and torch.equal(acc2.cpu(), host[i][0][0])
        assert torch.equal(me[0].cpu(), host[i][1][0]) and torch.equal(mp[0].cpu(), host[i][2][0])
    assert pf.bytes_copied > 0


def test_prefetcher_end_of_data_and_shape_changes():
    """fetch() raising StopIteration ends the stream after the last batch is delivered; batches of changing mask widths
    re-allocate their slot buffers."""
    from jepa_amd.engine.input import DevicePrefetcher
    host = [([torch.full((1, 3, 2, 8, 8), float(i))], [torch.arange(4 + i).view(1, -1)], [torch.arange(3).view(1, -1)])
            for i in range(3)]
    it = iter(host)
    pf = DevicePrefetcher(lambda: next(it), torch.device(DEV))
    for i in range(3):
        c, me, mp = pf.next()
        assert float(c.flatten()[0]) == float(i) and me[0].shape[1] == 4 + i
    with pytest.raises(StopIteration):
        pf.next()


# ------------------------------------------------------------------------------------------------ input edge
def test_prefetcher_pageable_inputs_with_the_host_running_ahead():
    """with pageable loader tensors the prefetcher stages through pinned buffers; a host that runs
    several steps ahead of the GPU (nothing reads the loss) must not overwrite a pinned buffer whose H2D copy has not
    executed yet.  The GPU is kept ~60 ms behind per step by a dummy load; every delivered batch must carry its own
    constant (checked after ONE final synchronise)."""
    from jepa_amd.engine.input import DevicePrefetcher
    n, shape = 8, (8, 3, 16, 112, 112)
    cnt = [0]

    def fetch():
        i = cnt[0]
        cnt[0] += 1
        if i >= n:
            raise StopIteration
        return ([torch.full(shape, float(i))], [torch.full((8, 5), i, dtype=torch.int64)],
                [torch.full((8, 3), i, dtype=torch.int64)])
    pf = DevicePrefetcher(fetch, torch.device(DEV))
    heavy = torch.randn(6144, 6144, device=DEV)
    seen = []
    for i in range(n):
        for _ in range(8):
            heavy @ heavy                       # test-side load only: keeps the device behind the host
        c, me, mp = pf.next()
        seen.append(torch.stack([c.min(), c.max(), me[0].max().float(), mp[0].min().float()]))
    torch.cuda.synchronize()
    for i, s in enumerate(seen):
        assert s.tolist() == [float(i)] * 4, (i, s.tolist())


# ------------------------------------------------------------------------------------------------ module forwards
def test_standalone_block_forwards_match_torch():
    """MLP / Attention / Block / PatchEmbed3D.forward (inference, no grad) vs the same arithmetic in fp32 torch:
    rel-L2 <= 1.5e-2 (bf16 operands)."""
    import torch.nn.functional as F
    from jepa_amd.src.models.utils.modules import Block
    from jepa_amd.src.models.utils.patch_embed import PatchEmbed3D
    torch.manual_seed(0)
    blk = Block(dim=128, num_heads=4, mlp_ratio=4.0, qkv_bias=True,
                norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6)).to(DEV)
    with torch.no_grad():
        for p in blk.parameters():
            p.add_(0.05 * torch.randn_like(p))
    x = torch.randn(3, 70, 128, device=DEV)

    def ref_attn(a, t):
        B, N, C = t.shape
        qkv = F.linear(t, a.qkv.weight, a.qkv.bias).reshape(B, N, 3, a.num_heads, C // a.num_heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, N, C)
        return F.linear(o, a.proj.weight, a.proj.bias)

    def ref_mlp(m, t):
        return F.linear(F.gelu(F.linear(t, m.fc1.weight, m.fc1.bias)), m.fc2.weight, m.fc2.bias)

    with torch.no_grad():
        r_attn, r_mlp = ref_attn(blk.attn, x), ref_mlp(blk.mlp, x)
        y = x + ref_attn(blk.attn, blk.norm1(x))
        r_blk = y + ref_mlp(blk.mlp, blk.norm2(y))
        assert rel_l2(blk.attn(x).cpu(), r_attn.cpu()) < 1.5e-2
        assert rel_l2(blk.mlp(x).cpu(), r_mlp.cpu()) < 1.5e-2
        assert rel_l2(blk(x).cpu(), r_blk.cpu()) < 1.5e-2
        pe = PatchEmbed3D(patch_size=16, tubelet_size=2, in_chans=3, embed_dim=64).to(DEV)
        clip = torch.randn(2, 3, 4, 32, 32, device=DEV)
        r_pe = pe.proj(clip).flatten(2).transpose(1, 2)
        assert rel_l2(pe(clip).cpu(), r_pe.cpu()) < 1.5e-2
    with pytest.raises(NotImplementedError):
        blk(x)                                            # grad mode: the stand-alone forward refuses
    with pytest.raises(ValueError):
        with torch.no_grad():
            blk.cpu()(x.cpu())                            # and there is no CPU path


def test_frozen_encoder_inference_matches_oracle():
    """The frozen-eval case (evals/video_classification_frozen/eval.py:414-441: every parameter requires_grad=False, called
    under no_grad): one C launch chain per forward with the two-workgroups-per-CU GEMM.  Output vs the fp32 oracle on the
    same weights: rel-L2 <= 2e-2; identical (bitwise) to the automatic GEMM selection."""
    from oracle import vjepa_oracle as O
    import jepa_amd.src.models.vision_transformer as V
    enc, _ = build_models(TINY, 2, perturb_small=True)
    vit = enc.backbone
    w = {k: v.detach().clone() for k, v in vit.state_dict().items()}
    for p in vit.parameters():
        p.requires_grad = False
    vit.to(DEV)
    clips = torch.randn(3, 3, TINY["frames"], TINY["crop"], TINY["crop"], generator=torch.Generator().manual_seed(7))
    ref = O.encoder_forward(w, clips, oracle_cfg(TINY, 2))
    with torch.no_grad():
        y = vit(clips.to(DEV))
        assert rel_l2(y.float().cpu(), ref) < 2e-2
        old = V.INFER_GEMM_FLAGS
        try:
            V.INFER_GEMM_FLAGS = 0
            y0 = vit(clips.to(DEV))
        finally:
            V.INFER_GEMM_FLAGS = old
        assert torch.equal(y, y0)
        idx = torch.stack([torch.randperm(TINY["num_patches"])[:20].sort().values for _ in range(3)]).to(DEV)
        ym = vit(clips.to(DEV), [idx])
        refm = O.encoder_forward(w, clips, oracle_cfg(TINY, 2), idx.cpu())
        assert rel_l2(ym.float().cpu(), refm) < 2e-2

