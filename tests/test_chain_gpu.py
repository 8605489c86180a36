"""The C launch chains (csrc/chain.hip: vj_blocks_fwd / vj_blocks_bwd; reference block loops src/models/vision_transformer.py:181-184,
predictor.py:231-232 and their autograd graph): bit-identical to the per-kernel Python chain, argument checks, the soft-max scale applied
by the qkv GEMM, bias gradients from the producing kernels, workspace guard bands, and the run-time options round trip."""
import os
import socket
import sys
import pytest
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import HP, rel_l2  # noqa: E402
from tests.step_util import (TINY, TINY_MASKS, build_models, build_trainer, draw_batch, oracle_cfg,  # noqa: E402
                             to_dev)
import math
from tests.golden_util import rel_l2  # noqa: E402
from tests.step_util import TINY, TINY_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402
from tests.gpu_util import ATTN_SHAPES, bf, sdpa_ref  # noqa: E402,F401
import ctypes
from tests.step_util import TINY, TINY_MASKS, VITH, VITL, VITL_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


DEV = "cuda"


def _gens(masks=TINY_MASKS, m=TINY):
    from oracle import vjepa_oracle as O
    return O.make_mask_gens(masks, m["crop"], m["frames"], m["patch"], m["tubelet"])


# ------------------------------------------------------------------------------------------ the chain with / without bias_fuse
def fused_bias_mask(tr, arena=None):
    """bool mask over a parameter arena (default: the trainable one): True on the qkv / fc1 biases (the only gradients option
    bias_fuse changes)."""
    arena = tr.arena if arena is None else arena
    lo = getattr(arena, "lo", 0)                 # the EMA target arena covers the encoder range [lo, hi) of the trainer arena
    m = torch.zeros(arena.P.numel(), dtype=torch.bool, device=arena.P.device)
    for name, sl in arena.slots.items():
        if name.endswith("attn.qkv.bias") or name.endswith("mlp.fc1.bias"):
            m[sl.off - lo:sl.off - lo + sl.numel] = True
    return m


@pytest.fixture(scope="module")
def ops():
    from jepa_amd.hip import ops as o
    return o


class _opt:
    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        from jepa_amd.hip.lib import set_option
        self.old = set_option(self.name, self.value)

    def __exit__(self, *a):
        from jepa_amd.hip.lib import set_option
        set_option(self.name, self.old)


# ------------------------------------------------------------------------------------------ guard bands
BAND = 4096


PATTERN = 0xA5


class _GuardedWorkspaces:
    """Replaces engine.chain.Workspace.get and hip.ops.Scratch.get by allocators that return EXACTLY the requested bytes from
    the middle of a buffer whose first and last 4 KB hold a byte pattern; check() asserts both bands of every buffer."""

    def __enter__(self):
        from jepa_amd.engine import chain
        from jepa_amd.hip import ops
        self.chain, self.ops = chain, ops
        self.bufs = {}
        self.n_gaps = 0
        self.old_ws, self.old_sc = chain.Workspace.get, ops.Scratch.get

        def alloc(key, nbytes, device):
            nbytes = (int(nbytes) + 255) // 256 * 256
            ent = self.bufs.get(key)
            if ent is None or ent[1] != nbytes:
                torch.cuda.synchronize()
                if ent is not None:   # the buffer about to be replaced: its bands and the gaps recorded inside it must be intact
                    assert bool((ent[0][:BAND] == PATTERN).all()) and bool((ent[0][BAND + ent[1]:] == PATTERN).all()), key
                    n, bad = _guard_check()   # (inspects and forgets every recorded gap: none may point into freed memory later)
                    self.n_gaps += n
                    assert bad == 0, (key, n, bad)
                raw = torch.empty(nbytes + 2 * BAND, dtype=torch.uint8, device=device)
                raw[:BAND] = PATTERN
                raw[BAND + nbytes:] = PATTERN
                ent = self.bufs[key] = (raw, nbytes)
            return ent[0][BAND:BAND + nbytes]

        def ws_get(tag, nbytes, device):
            return alloc(("ws", tag), nbytes, device)

        def sc_get(nbytes, device, tag="default", stream=None):
            return alloc(("sc", tag, ops._raw_stream(torch.cuda.current_device()) if stream is None else stream), max(int(nbytes), 1 << 20), device)
        chain.Workspace.get = staticmethod(ws_get)
        ops.Scratch.get = staticmethod(sc_get)
        return self

    def check(self, what):
        torch.cuda.synchronize()
        for key, (raw, nbytes) in self.bufs.items():
            head, tail = raw[:BAND], raw[BAND + nbytes:]
            assert bool((head == PATTERN).all()), (what, key, "band BEFORE the workspace was written")
            assert bool((tail == PATTERN).all()), (what, key, "band AFTER the workspace was written")

    def __exit__(self, *a):
        self.chain.Workspace.get, self.ops.Scratch.get = self.old_ws, self.old_sc
        torch.cuda.synchronize()
        try:
            _guard_check()   # forget gaps that point into the buffers released below
        finally:
            self.bufs.clear()


def _guard_check():
    from jepa_amd.hip.lib import check, load_library
    n, bad = ctypes.c_int64(0), ctypes.c_int64(0)
    check(load_library().vj_ws_guard_check(ctypes.byref(n), ctypes.byref(bad)), "vj_ws_guard_check")
    return n.value, bad.value


# (model, masks, batch, micro-batch, tile orders): the BASELINE model sizes at batches that keep the test in seconds; the benched
# ViT-L batch itself with the default order and the two other families of orders
GUARD_CASES = [
    # (round 5 also ran ViT-L B = 24 and ViT-H B = 6 in micro-batches of 3: > 3000 gaps, none damaged -- profiles/r05_guard_bands.md)
    ("vitl_b4_orders", VITL, VITL_MASKS, 4, None,
     (0, 1, 4, 8, 33, 255, 256, 258, 260, 262, 264, 300, 384, 511)),   # (all 28 orders of round 5 ran clean for two rounds; half of them kept)
    ("vith384_b1", dict(VITH, crop=384, num_patches=4608), VITL_MASKS, 1, None, (260,)),
    ("tiny_b2", TINY, TINY_MASKS[:1], 2, None, (260,)),
]


# ------------------------------------------------------------------------------------------------ launch chains
def test_c_chain_is_bit_identical_to_python_chain():
    """vj_blocks_fwd / vj_blocks_bwd enqueue the same kernels in the same order as the per-kernel Python chain:
    losses, every gradient and every updated weight must be BIT-identical (split-K is deterministic)."""
    from jepa_amd.engine import layers
    from jepa_amd.hip.lib import set_option
    res = {}
    old_bf = set_option("bias_fuse", 0)   # the Python chain has no fused bias route; bias_fuse 1 vs 0 is covered below
    for use_c in (True, False):
        layers.USE_C_CHAIN = use_c
        gens = _gens()    # fresh generators: their step counters seed the block sizes (multiblock3d.py:114-128)
        try:
            tr, _, _, _, _ = build_trainer(TINY, 2)
            for step in range(1, 3):
                clips, me, mp = draw_batch(gens, 4, TINY, 10 + step, 20 + step)
                out = tr.train_step(*to_dev(clips, me, mp), lr=1e-3, wd=0.04, ema=0.99)
            res[use_c] = (out.loss, tr.arena.G.clone(), tr.arena.P.clone(), tr.tarena.P.clone())
        finally:
            layers.USE_C_CHAIN = True
            if not use_c or use_c not in res:
                set_option("bias_fuse", old_bf)
    assert res[True][0] == res[False][0]
    # round 4: the C chain takes the qkv / fc1 bias gradients from column partials of the kernels that produce dqkv / du (option
    # bias_fuse, fp32 sums of the un-rounded values), the per-kernel Python chain from the stand-alone column sums of the bf16
    # dY: those tensors (and, after AdamW, their weights) agree to rounding, everything else stays bit-identical
    from tests.gpu_util import fused_bias_mask
    m = fused_bias_mask(tr)
    for k, (a, b) in enumerate(zip(res[True][1:3], res[False][1:3])):
        assert torch.equal(a[~m], b[~m]), k
        assert rel_l2(a[m].cpu(), b[m].cpu()) < 3e-3, (k, rel_l2(a[m].cpu(), b[m].cpu()))
    mt = fused_bias_mask(tr, tr.tarena)          # the EMA target follows the encoder's biases
    assert torch.equal(res[True][3][~mt], res[False][3][~mt])
    assert rel_l2(res[True][3][mt].cpu(), res[False][3][mt].cpu()) < 1e-4


# ------------------------------------------------------------------------------------------------ edge cases
def test_chain_rejects_bad_arguments_before_launching():
    """vj_blocks_fwd / vj_blocks_bwd validate on the host: too-small or misaligned workspaces, segments that do not tile
    the rows, inconsistent Linear shapes -> negative return code + vj_last_error(), nothing enqueued."""
    import ctypes
    from jepa_amd.engine import chain
    from jepa_amd.engine.layers import Seg
    from jepa_amd.hip.lib import load_library
    lib = load_library()
    tr, _, _, _, _ = build_trainer(TINY, 2)
    ew = tr.ew
    M, D = 96, 192
    x = torch.randn(M, D, device=DEV).bfloat16()
    out = torch.empty_like(x)
    arr = chain.block_array(ew.blocks)
    n = len(ew.blocks)
    st = torch.cuda.current_stream().cuda_stream
    need = lib.vj_blocks_fwd_ws_bytes(M, D, 4 * D, ew.heads, n, 1)
    ws = torch.empty(need + 512, dtype=torch.uint8, device=DEV)
    good = chain.seg_array([Seg(0, 2, 48)])

    def fwd(segs, nseg, ws_ptr, ws_bytes, heads=ew.heads):
        return lib.vj_blocks_fwd(arr, n, x.data_ptr(), out.data_ptr(), M, D, heads, segs, nseg, 1e-6, 1, 0, ws_ptr, ws_bytes, st)
    assert fwd(good, 1, ws.data_ptr(), ws.numel()) == 0
    assert fwd(good, 1, ws.data_ptr(), need - 256) < 0 and b"workspace too small" in lib.vj_last_error()
    assert fwd(good, 1, ws.data_ptr() + 64, ws.numel() - 64) < 0 and b"aligned" in lib.vj_last_error()
    assert fwd(chain.seg_array([Seg(0, 2, 40)]), 1, ws.data_ptr(), ws.numel()) < 0 and b"segments" in lib.vj_last_error()
    assert fwd(chain.seg_array([Seg(8, 2, 44)]), 1, ws.data_ptr(), ws.numel()) < 0
    assert fwd(good, 1, ws.data_ptr(), ws.numel(), heads=5) < 0 and b"divisible" in lib.vj_last_error()
    # backward without transposed weights / gradient views (a forward-only descriptor) is refused
    tw_arr = chain.block_array(tr.tw.blocks)
    nb = lib.vj_blocks_bwd_ws_bytes(M, D, 4 * D, ew.heads)
    tmp = torch.empty(nb, dtype=torch.uint8, device=DEV)
    rc = lib.vj_blocks_bwd(tw_arr, n, x.data_ptr(), out.data_ptr(), out.data_ptr(), M, D, ew.heads, good, 1, 1.0, 0.0,
                           ws.data_ptr(), ws.numel(), tmp.data_ptr(), tmp.numel(), 0, st, None, chain.LAYER_CB(), None)
    assert rc < 0 and b"lacks transposed weights" in lib.vj_last_error()
    torch.cuda.synchronize()


def test_block_chain_with_the_scale_in_the_qkv_gemm():
    """Option attn_softmax = 2 in the block chains (qkv GEMM epilogue 4 + attention told "q is pre-scaled"): the C chain and the
    per-kernel Python chain stay bit-identical, and the step agrees with option 1 (scale folded inside the attention kernels) to
    bf16 noise: loss 2e-4 relative, gradient arena 2e-2 rel-L2."""
    from jepa_amd.engine import layers
    tr, _, _, _, _ = build_trainer(TINY, 2, perturb_small=True)
    clips, me, mp = draw_batch(_gens(), 4, TINY, 71, 72)
    cd, med, mpd = to_dev(clips, me, mp)
    res = {}
    with _opt("bias_fuse", 0):
        for key, (sm, c_chain) in {"c2": (2, True), "py2": (2, False), "c1": (1, True)}.items():
            layers.USE_C_CHAIN = c_chain
            try:
                with _opt("attn_softmax", sm):
                    o = tr.train_step(cd, med, mpd, lr=0.0, wd=0.0, ema=1.0)
                    torch.cuda.synchronize()
                res[key] = (o.loss, tr.arena.G.clone())
            finally:
                layers.USE_C_CHAIN = True
    assert res["c2"][0] == res["py2"][0] and torch.equal(res["c2"][1], res["py2"][1])
    assert abs(res["c2"][0] - res["c1"][0]) <= 2e-4 * abs(res["c1"][0]), (res["c2"][0], res["c1"][0])
    r = rel_l2(res["c2"][1].cpu(), res["c1"][1].cpu())
    assert r < 2e-2, r


def test_block_chain_bias_fuse_changes_only_the_fused_biases():
    """One step on the same weights / batch with option bias_fuse = 1 and 0: the loss and every gradient except the qkv / fc1
    biases are BIT-identical (the segmented reduction keeps the summation order of the single reductions; the last block's fc2
    bias comes out of the final-norm backward in both), and the fused biases -- fp32 sums of the un-rounded dY instead of sums
    of the bf16 dY -- agree to 3e-3 rel-L2 per tensor.  TINY model: sequences of 16-48 tokens (partial-row bound, GEMMs below
    the persistent kernel's size -> the fc1 bias silently takes the unfused route: both must stay correct)."""
    tr, _, _, _, _ = build_trainer(TINY, 2, perturb_small=True)
    clips, me, mp = draw_batch(_gens(), 4, TINY, 61, 62)
    cd, med, mpd = to_dev(clips, me, mp)
    res = {}
    for bfz in (1, 0):
        with _opt("bias_fuse", bfz):
            o = tr.train_step(cd, med, mpd, lr=0.0, wd=0.0, ema=1.0)
            torch.cuda.synchronize()
        res[bfz] = (o.loss, tr.arena.G.clone())
    assert res[1][0] == res[0][0]
    m = fused_bias_mask(tr)
    assert torch.equal(res[1][1][~m], res[0][1][~m])
    for name, sl in tr.arena.slots.items():
        if name.endswith("attn.qkv.bias") or name.endswith("mlp.fc1.bias"):
            a, b = res[1][1][sl.off:sl.off + sl.numel], res[0][1][sl.off:sl.off + sl.numel]
            assert rel_l2(a.cpu(), b.cpu()) < 3e-3, (name, rel_l2(a.cpu(), b.cpu()))


@pytest.mark.timeout(900)
def test_block_chain_bias_fuse_vitl_b24_and_micro_batches():
    """The same at the benched size (ViT-L/16 16x224x224, B = 24: every fused route active -- attention partials for sequences
    of 48-1248 tokens, the fc2-dgrad epilogue sums on the persistent kernel incl. shifted last row tiles), run-to-run bitwise
    determinism of the fused path, and gradient accumulation over micro-batches of 12 through the fused path (beta = 1 on the
    one reduction launch): arena rel-L2 <= 2e-5 against the full batch."""
    from oracle import vjepa_oracle as O
    from tests.step_util import VITL, VITL_MASKS
    tr, _, _, _, _ = build_trainer(VITL, 2)
    gens = O.make_mask_gens(VITL_MASKS, VITL["crop"], VITL["frames"], VITL["patch"], VITL["tubelet"])
    clips, me, mp = draw_batch(gens, 24, VITL, 1234, 4321)
    cd, med, mpd = to_dev(clips, me, mp)

    def run(bfz, mb=None):
        tr.micro_batch = mb
        try:
            with _opt("bias_fuse", bfz):
                o = tr.train_step(cd, med, mpd, lr=0.0, wd=0.0, ema=1.0)
                torch.cuda.synchronize()
            return o.loss, tr.arena.G.clone()
        finally:
            tr.micro_batch = None
    l1, g1 = run(1)
    l1b, g1b = run(1)
    assert l1 == l1b and torch.equal(g1, g1b), "the fused path is not deterministic run-to-run"
    l0, g0 = run(0)
    assert l1 == l0
    m = fused_bias_mask(tr)
    assert torch.equal(g1[~m], g0[~m])
    worst = 0.0
    for name, sl in tr.arena.slots.items():
        if name.endswith("attn.qkv.bias") or name.endswith("mlp.fc1.bias"):
            e = rel_l2(g1[sl.off:sl.off + sl.numel].cpu(), g0[sl.off:sl.off + sl.numel].cpu())
            worst = max(worst, e)
            assert e < 3e-3, (name, e)
    print(f"bias_fuse 1 vs 0 at ViT-L B=24: worst fused-bias rel-L2 {worst:.2e}")
    lm, gm = run(1, mb=12)
    assert abs(lm - l1) <= 1e-6 * abs(l1)
    r = float((gm.double() - g1.double()).norm() / g1.double().norm())
    assert r < 2e-5, r


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name,model,masks,B,micro,orders", GUARD_CASES, ids=[c[0] for c in GUARD_CASES])
def test_chain_workspaces_stay_inside_their_guard_bands(name, model, masks, B, micro, orders):
    """Two training steps (the second with different mask sizes, i.e. other sequence lengths in the same buffers) per tile order:
    no 256-byte gap behind a workspace member (saved activations, backward temporaries, LayerNorm / attention / fc2-dgrad column
    partials, split-K partials) and no 4 KB band around a workspace or a scratch buffer may change."""
    from oracle import vjepa_oracle as O
    with _opt("ws_guard", 1), _GuardedWorkspaces() as gw:
        tr, _, _, _, _ = build_trainer(model, len(masks), micro_batch=micro, overlap_update=True)
        gens = O.make_mask_gens(masks, model["crop"], model["frames"], model["patch"], model["tubelet"])
        batches = [to_dev(*draw_batch(gens, B, model, 700 + i, 800 + i)) for i in range(2)]
        for raster in orders:
            with _opt("gemm_raster", raster):
                for cd, med, mpd in batches:
                    o = tr.train_step(cd, med, mpd, lr=1e-4, wd=0.04, ema=0.998)
                tr.sync_update()
                assert 0.05 < o.loss < 5.0 and not o.skipped, (name, raster, o.loss)
                n, bad = _guard_check()
                assert bad == 0, (name, raster, n, bad)
                gw.n_gaps += n
                gw.check((name, raster))
        assert gw.n_gaps > 100 * len(orders), gw.n_gaps
        _guard_check()
        del tr


# ------------------------------------------------------------------------------------------------ run-time options
def test_runtime_options_round_trip_and_unknown_name():
    from jepa_amd.hip.lib import HipKernelError, get_option, set_option
    old = set_option("gemm_4w", 1)
    try:
        assert get_option("gemm_4w") == 1
    finally:
        set_option("gemm_4w", old)
    with pytest.raises(HipKernelError):
        get_option("no_such_option")
    # round 6: twelve switches; the measured-negative forms of rounds 3-5 are gone with their options, and value sets are enforced
    for gone in ("gemm_sched", "gemm_dyn", "gemm_nt", "attn_dkdv_kt", "attn_dq_qw", "gelu_poly", "attn_psum", "attn_merge", "ln_bwd_prefetch",
                 "adam_grid", "wgrad_slow_issue"):
        with pytest.raises(HipKernelError):
            get_option(gone)
    for name, bad in (("gemm_epi_pre", 2), ("attn_softmax", 0), ("gemm_4w", 3)):
        with pytest.raises(HipKernelError):
            set_option(name, bad)

