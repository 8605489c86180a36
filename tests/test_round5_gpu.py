"""Round-5 GPU tests.

* the fused AdamW / EMA update issued range by range on its own stream (Trainer(overlap_update=True)) and the gated, range-wise
  enqueue of the trunks that goes with it: bitwise the same weights, moments, EMA target and shadows as the single-stream update;
* guard bands (one unexplained "Memory access fault" in round 4): every member of the chain workspaces
  followed by a poisoned 256-byte gap (option ws_guard) + poisoned 4 KB bands around the workspaces themselves, over the
  BASELINE model sizes and over tile orders of the persistent GEMM; single GEMMs with poisoned bands around every output for
  ALL 512 values option gemm_raster admits.
Every kernel is reached through the C ABI; tolerances (here: bitwise) are next to each assertion."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.step_util import TINY, TINY_MASKS, VITH, VITL, VITL_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from jepa_amd.hip import ops as o
    return o


def _gens(masks=TINY_MASKS, m=TINY):
    from oracle import vjepa_oracle as O
    return O.make_mask_gens(masks, m["crop"], m["frames"], m["patch"], m["tubelet"])


class _opt:
    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        from jepa_amd.hip.lib import set_option
        self.old = set_option(self.name, self.value)

    def __exit__(self, *a):
        from jepa_amd.hip.lib import set_option
        set_option(self.name, self.old)


# ------------------------------------------------------------------------------------------ deferred, range-wise update
def _state(tr):
    tr.sync_update()
    torch.cuda.synchronize()
    A, T = tr.arena, tr.tarena
    out = dict(P=A.P.clone(), Pb=A.Pb.clone(), M1=A.M1.clone(), M2=A.M2.clone(), G=A.G.clone(), TP=T.P.clone(), TPb=T.Pb.clone())
    for n, t in A.wT.items():
        out["wT:" + n] = t.clone()
    return out


@pytest.mark.parametrize("micro", [None, 2])
def test_overlapped_update_is_bit_identical(micro):
    """ViT-Tiny (12 blocks -> four encoder ranges + the predictor), six steps with warm-up-style lr, weight decay, EMA < 1 and an
    active clip: Trainer(overlap_update=True) -- update on its own stream, the next step's forwards gated range by range, the
    trunks enqueued as one vj_blocks_fwd call per range -- against the plain Trainer.  Everything the step owns is compared
    bitwise after every step: master weights, bf16 / transposed shadows, both Adam moments, gradients, EMA target."""
    trs = [build_trainer(TINY, 2, perturb_small=True, clip_grad=0.05, micro_batch=micro, overlap_update=ov)[0] for ov in (False, True)]
    plan = trs[1]._plan
    assert [w for w, _, _ in plan] == ["enc"] * 4 + ["pred"] and [f for _, f, _ in plan][:4] == [0, 1, 3, 6], plan
    # the ranges tile the four groups exactly
    for gi in range(4):
        lo, hi = trs[1].arena.group_ranges[gi]
        segs = sorted((a, b) for _, _, rs in plan for g, a, b in rs if g == gi)
        assert segs[0][0] == lo and segs[-1][1] == hi and all(x[1] == y[0] for x, y in zip(segs, segs[1:])), (gi, segs)
    losses = [[], []]
    for step in range(6):
        gens = _gens()
        for _ in range(step + 1):
            clips, me, mp = draw_batch(gens, 4, TINY, 300 + step, 400 + step)
        cd, med, mpd = to_dev(clips, me, mp)
        for k, tr in enumerate(trs):
            o = tr.train_step(cd, med, mpd, lr=1e-3 * (step + 1), wd=0.04, ema=0.99, clip_now=step >= 2)
            losses[k].append(o)
        a, b = _state(trs[0]), _state(trs[1])
        for key in a:
            assert torch.equal(a[key], b[key]), (step, key, int((a[key] != b[key]).sum()))
    for oa, ob in zip(*losses):
        assert oa.loss == ob.loss and oa.raw_grad_norms == ob.raw_grad_norms and not ob.skipped


def test_module_forward_waits_for_a_pending_update():
    """Readers inside the package order themselves against a deferred update: the module-level encoder forward right after a
    train_step of an overlap_update Trainer sees the UPDATED weights (compared with the same call after a full synchronise)."""
    tr, _, enc, _, _ = build_trainer(TINY, 2, overlap_update=True)
    clips, me, mp = draw_batch(_gens(), 4, TINY, 11, 12)
    cd, med, mpd = to_dev(clips, me, mp)
    tr.train_step(cd, med, mpd, lr=1e-2, wd=0.0, ema=0.9)
    with torch.no_grad():
        z_now = [t.clone() for t in enc(cd, med)]
        torch.cuda.synchronize()
        tr.sync_update()
        z_later = enc(cd, med)
    for a, b in zip(z_now, z_later):
        assert torch.equal(a, b)


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
    ("vitl_b24", VITL, VITL_MASKS, 24, None, (260, 0, 262)),
    ("vitl_b4_orders", VITL, VITL_MASKS, 4, None,
     (0, 1, 2, 3, 4, 5, 7, 8, 16, 33, 64, 128, 255, 256, 257, 258, 259, 260, 261, 262, 264, 272, 300, 383, 384, 400, 510, 511)),
    ("vith_b6_micro3", VITH, VITL_MASKS, 6, 3, (260, 8, 511)),
    ("vith384_b1", dict(VITH, crop=384, num_patches=4608), VITL_MASKS, 1, None, (260, 0)),
    ("tiny_b2", TINY, TINY_MASKS[:1], 2, None, (260,)),
]


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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("M,N,K", [(5000, 1288, 256), (2304 + 40, 2600, 320)])
def test_every_tile_order_of_the_persistent_gemm_inside_guard_bands(ops, M, N, K):
    """ALL 512 values of option gemm_raster (csrc/options.cpp admits 0 ... 511) on shapes with >= 90 tiles (the persistent kernel's
    threshold) whose last row AND column tile are shifted: outputs in the middle of poisoned buffers (plain, residual, GELU + saved derivative, fc2-dgrad + column partials);
    every order must give the bits of order 0 and leave the 4 KB bands on both sides of every output untouched."""
    g = torch.Generator(device=DEV).manual_seed(43)
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
    aux = (torch.rand(M, N, device=DEV, generator=g) * 1.2 - 0.1).to(torch.bfloat16)
    nb = M * N * 2

    def banded():
        raw = torch.full((nb + 2 * BAND,), PATTERN, dtype=torch.uint8, device=DEV)
        return raw, raw[BAND:BAND + nb].view(torch.bfloat16).view(M, N)

    def run_all():
        raws, outs = [], []
        for kw in (dict(bias=bias), dict(bias=bias, residual=res), dict(bias=bias, epilogue=ops.EPI_GELU)):
            raw, out = banded()
            if kw.get("epilogue") == ops.EPI_GELU:
                raw2, out2 = banded()
                kw["aux_out"] = out2
                raws.append(raw2)
                outs.append(out2)
            ops.gemm_nt(A, W, out=out, **kw)
            raws.append(raw)
            outs.append(out)
        du, colpart = ops.gemm_dgelu_colsum(A, W, aux)
        outs.append(du)
        if colpart is not None:
            outs.append(colpart)
        torch.cuda.synchronize()
        for raw in raws:
            assert bool((raw[:BAND] == PATTERN).all()) and bool((raw[BAND + nb:] == PATTERN).all())
        return outs
    with _opt("gemm_raster", 0):
        ref = run_all()
    for raster in range(1, 512):
        with _opt("gemm_raster", raster):
            got = run_all()
        for i, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), (raster, i, int((a != b).sum()))


# ------------------------------------------------------------------------------------------ persistent two-workgroups-per-CU GEMM
@pytest.mark.parametrize("M,N,K", [(5000, 1288, 256), (10560, 1024, 1024), (2304 + 40, 2600, 320), (37632, 384, 384), (1024, 384, 1536),
                                   (256, 128, 192)])
def test_persistent_4wave_gemm_is_bit_identical(ops, M, N, K):
    """gemm_nt_4wp_kernel (flags 0x200: 256 x 128 tiles, 4 waves, two workgroups per CU walking tile lists, edge tiles shifted inside
    the matrix) against the automatic selection -- same K loop arithmetic and epilogues, hence the same bits: plain + bias,
    residual, GELU with and without the saved derivative, dGELU."""
    g = torch.Generator(device=DEV).manual_seed(47)
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
    aux = (torch.rand(M, N, device=DEV, generator=g) * 1.2 - 0.1).to(torch.bfloat16)

    def run_all(flags):
        outs = [ops.gemm_nt(A, W, bias=bias, flags=flags), ops.gemm_nt(A, W, bias=bias, residual=res, flags=flags),
                ops.gemm_nt(A, W, bias=bias, epilogue=ops.EPI_GELU, flags=flags)]
        u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        outs.append(ops.gemm_nt(A, W, bias=bias, aux_out=u, epilogue=ops.EPI_GELU, flags=flags))
        outs.append(u)
        outs.append(ops.gemm_nt(A, W, aux_in=aux, epilogue=ops.EPI_DGELU, flags=flags))
        torch.cuda.synchronize()
        return outs
    got = run_all(0x200)
    refs = [run_all(0x100)]                 # the one-tile 4-wave kernel: the same K loop, any shape
    if M * N >= 90 * 65536:
        refs.append(run_all(0))             # the automatic selection = the persistent 8-phase kernel at >= 90 tiles of 256 x 256
    for ref in refs:
        for i, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), (i, int((a != b).sum()))


# ------------------------------------------------------------------------------------------ attention backward, larger per-wave tiles
@pytest.mark.parametrize("B,S,H,hd", [(3, 1232, 16, 24), (2, 300, 4, 24), (2, 77, 3, 32), (1, 513, 2, 16), (2, 1152, 12, 24)])
@pytest.mark.parametrize("scale_sign", [1.0, -1.0])
def test_attention_backward_with_four_tiles_per_wave_is_bit_identical(ops, B, S, H, hd, scale_sign):
    """Options attn_dkdv_kt = 4 (64 keys per wave in dK/dV) and attn_dq_qw = 4 (64 queries per wave in dQ) at head_dim <= 32: each
    key's dK / dV and each query's dQ go through the same operations in the same order whatever the workgroup partition, so dqkv
    is BIT-identical to the default tiling (which round 4 checks against fp32 SDPA); the qkv-bias column partials regroup (other
    partial rows), their sums agree to fp32 rounding.  Positive scale and the pre-scaled-q convention (negative scale)."""
    g = torch.Generator(device=DEV).manual_seed(53)
    qkv = torch.randn(B * S, 3 * H * hd, device=DEV, generator=g).to(torch.bfloat16)
    dout = torch.randn(B * S, H * hd, device=DEV, generator=g).to(torch.bfloat16)
    scale = scale_sign * hd ** -0.5
    o, lse = ops.attn_fwd(qkv, B, S, H, hd, scale)
    ref, cq0, ckv0 = ops.attn_bwd_colsum(qkv, o, dout, lse, B, S, H, hd, scale)
    torch.cuda.synchronize()
    for kt, qw in ((4, 0), (0, 4), (4, 4)):
        with _opt("attn_dkdv_kt", kt), _opt("attn_dq_qw", qw):
            got, cq, ckv = ops.attn_bwd_colsum(qkv, o, dout, lse, B, S, H, hd, scale)
            plain = ops.attn_bwd(qkv, o, dout, lse, B, S, H, hd, scale)
            torch.cuda.synchronize()
        assert torch.equal(got, ref), (kt, qw, int((got != ref).sum()))
        assert torch.equal(plain, ref), (kt, qw)
        for a, b in ((cq, cq0), (ckv, ckv0)):
            sa, sb = a.sum(0).double(), b.sum(0).double()
            assert float((sa - sb).norm() / (sb.norm() + 1e-30)) < 1e-5, (kt, qw)


# ------------------------------------------------------------------------------------------ independent streams
def test_independent_stream_picker():
    """engine.layers.independent_stream returns a stream whose kernels run concurrently with those of the main and of the side
    stream (vj_probe_spin on both, wall clock); a stream is never 'concurrent' with itself."""
    from jepa_amd.engine import layers
    main_s = torch.cuda.current_stream()
    side = layers.side_stream(torch.device(DEV)).stream
    assert layers.streams_concurrent(main_s, side)
    assert not layers.streams_concurrent(side, side)
    for _ in range(3):
        s = layers.independent_stream(torch.device(DEV), [main_s, side])
        assert layers.streams_concurrent(s, main_s) and layers.streams_concurrent(s, side)


# ------------------------------------------------------------------------------------------ LayerNorm folded into the consuming GEMM
def _ln_case(M, K, N, seed, adversarial):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(M, K, device=DEV, generator=g) * (1.0 + 2.0 * torch.rand(M, 1, device=DEV, generator=g))
    if adversarial:          # rows whose mean dwarfs their spread (|mean| up to 60 sigma), and a few huge-variance rows
        x = x + torch.randn(M, 1, device=DEV, generator=g) * 60.0
        x[::7] *= 30.0
    else:
        x = x + torch.randn(M, 1, device=DEV, generator=g) * 0.5
    x = x.to(torch.bfloat16)
    W = torch.randn(N, K, device=DEV, generator=g) * 0.03
    b = torch.randn(N, device=DEV, generator=g) * 0.1
    gamma = 1.0 + 0.3 * torch.randn(K, device=DEV, generator=g)
    beta = 0.2 * torch.randn(K, device=DEV, generator=g)
    return x, W, b, gamma, beta


@pytest.mark.parametrize("M,K,N", [(4096, 1024, 3072), (2304 + 40, 1024, 4096), (300, 192, 576), (77, 96, 384), (1000, 384, 1152)])
@pytest.mark.parametrize("adversarial", [False, True])
def test_layernorm_folded_into_the_gemm(ops, M, K, N, adversarial):
    """vj_ln_rowstats + vj_ln_fold_weights + vj_gemm_bf16_nt_lnfold against LayerNorm(x) W^T + b in float64 (from the same bf16 x):
      * the row statistics are those of layernorm_fwd_kernel bit for bit; Wf = bf16(W gamma) exactly, c / b' to fp32 rounding;
      * plain / q-scaled / GELU epilogues within the GEMM bound of the unfused path (rel-L2 4e-3) and never worse than 1.25x the
        unfused HIP path (LayerNorm kernel -> bf16 -> GEMM), which rounds the activation once more;
      * adversarial rows (|mean| ~ 60 sigma, 30x scale outliers): the epilogue's acc - mean * c cancels what the matrix pipe
        accumulated of the row mean -- exact up to fp32 accumulation, so the bound holds there too (the value to watch is stated)."""
    eps = 1e-6
    x, W, b, gamma, beta = _ln_case(M, K, N, 61 + M, adversarial)
    y_un, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps, save_stats=True)
    rs = ops.ln_rowstats(x, eps)
    assert torch.equal(rs[:, 0], rstd) and torch.equal(rs[:, 1], -mean * rstd)
    Wf, c, bf_ = ops.ln_fold_weights(W, b, gamma, beta)
    assert torch.equal(Wf, (W * gamma).to(torch.bfloat16))
    assert float((c.double() - Wf.double().sum(1)).abs().max()) < 1e-4 * float(Wf.double().abs().sum(1).max())
    assert float((bf_.double() - (b.double() + W.double() @ beta.double())).abs().max()) < 1e-5
    xd = x.double()
    ln = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + eps) * gamma.double() + beta.double()
    ref = ln @ W.double().t() + b.double()
    Wb = W.to(torch.bfloat16)
    hd_scale = 0.125 * 1.4426950408889634
    cases = [("plain", dict(epilogue=ops.EPI_BF16), ref, ops.gemm_nt(y_un, Wb, bias=b))]
    if N % 12 == 0:
        rq = ref.clone()
        rq[:, :N // 3] *= hd_scale
        cases.append(("qkv", dict(epilogue=ops.EPI_QKV, alpha=hd_scale), rq, ops.gemm_nt(y_un, Wb, bias=b, epilogue=ops.EPI_QKV, alpha=hd_scale)))
    cases.append(("gelu", dict(epilogue=ops.EPI_GELU), torch.nn.functional.gelu(ref), ops.gemm_nt(y_un, Wb, bias=b, epilogue=ops.EPI_GELU)))
    for name, kw, r, unfused in cases:
        out = ops.gemm_nt_lnfold(x, Wf, bf_, rs, c, **kw)
        torch.cuda.synchronize()
        e_f = float((out.double() - r).norm() / r.norm())
        e_u = float((unfused.double() - r).norm() / r.norm())
        print(f"[ln-fold {M}x{K}x{N} {'adv' if adversarial else 'std'} {name}] rel-L2 folded {e_f:.2e} | unfused {e_u:.2e}")
        assert e_f < 4e-3, (name, e_f, e_u)
        assert e_f < 1.25 * e_u + 1e-4, (name, e_f, e_u)


def test_target_forward_with_folded_layernorms():
    """Trainer.forward_target with and without the fold on the same weights (ViT-Tiny 12 blocks, perturbed affine parameters):
    both are bf16 evaluations of the same function -- rel-L2 between them at the bf16 level (2e-2 bound as against the oracle),
    each within 2e-2 of the fp32 oracle, and the C chain bit-identical to the per-kernel Python chain in both modes."""
    from jepa_amd.engine import layers
    from oracle import vjepa_oracle as O
    from tests.golden_util import rel_l2
    from tests.step_util import oracle_cfg
    tr, state, _, _, _ = build_trainer(TINY, 2, perturb_small=True)
    clips, me, mp = draw_batch(_gens(), 4, TINY, 21, 22)
    cd, med, mpd = to_dev(clips, me, mp)
    hs = {}
    for fold in (True, False):
        tr.set_ln_fold(fold)
        for c_chain in (True, False):
            layers.USE_C_CHAIN = c_chain
            try:
                hs[(fold, c_chain)] = [t.clone() for t in tr.forward_target(cd, mpd)]
            finally:
                layers.USE_C_CHAIN = True
        for a, b in zip(hs[(fold, True)], hs[(fold, False)]):
            assert torch.equal(a, b), ("C chain vs Python chain", fold)
    import torch.nn.functional as F
    with torch.no_grad():
        h = O.encoder_forward(state["tgt"], clips, oracle_cfg(TINY, 2))
        h = F.layer_norm(h, (TINY["embed_dim"],))
        ref = [O.take_rows(h, m) for m in mp]
    for i in range(len(mp)):
        e_fold = rel_l2(hs[(True, True)][i].float().cpu().reshape(ref[i].shape), ref[i])
        e_plain = rel_l2(hs[(False, True)][i].float().cpu().reshape(ref[i].shape), ref[i])
        e_between = rel_l2(hs[(True, True)][i].float().cpu(), hs[(False, True)][i].float().cpu())
        print(f"[target fold] mask {i}: vs fp32 oracle folded {e_fold:.2e} | unfused {e_plain:.2e} | folded vs unfused {e_between:.2e}")
        assert e_fold < 2e-2 and e_plain < 2e-2 and e_between < 2e-2


# ------------------------------------------------------------------------------------------ dynamic tile hand-out of the persistent GEMM
@pytest.mark.timeout(300)
@pytest.mark.parametrize("M,N,K", [(37632, 1024, 256), (5000, 1288, 256), (10560, 3072, 1088), (2304 + 40, 2600, 320), (58560, 384, 384)])
def test_dynamic_tile_handout_is_bit_identical(ops, M, N, K):
    """Option gemm_dyn = 1: tiles beyond a workgroup's first two come from per-XCD atomic counters (gemm8p.hip) instead of the static
    round-robin lists.  Another assignment of the same tiles to workgroups: the bits of the static lists, for every epilogue, with the
    trimmed and with the full grid -- and still after 1100 back-to-back launches (the 1024 counter slots wrap; each launch's last
    workgroup must have zeroed its slot)."""
    g = torch.Generator(device=DEV).manual_seed(59)
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
    aux = (torch.rand(M, N, device=DEV, generator=g) * 1.2 - 0.1).to(torch.bfloat16)

    def run_all():
        outs = [ops.gemm_nt(A, W, bias=bias), ops.gemm_nt(A, W, bias=bias, residual=res), ops.gemm_nt(A, W, bias=bias, epilogue=ops.EPI_GELU)]
        u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        outs.append(ops.gemm_nt(A, W, bias=bias, aux_out=u, epilogue=ops.EPI_GELU))
        outs.append(u)
        du, colpart = ops.gemm_dgelu_colsum(A, W, aux)
        outs.append(du)
        if colpart is not None:
            outs.append(colpart)
        torch.cuda.synchronize()
        return outs
    with _opt("gemm_dyn", 0):
        ref = run_all()
    for persist in (1, 2):
        with _opt("gemm_dyn", 1), _opt("gemm_persist", persist):
            for rep in range(3):
                got = run_all()
                for i, (a, b) in enumerate(zip(ref, got)):
                    assert torch.equal(a, b), (persist, rep, i, int((a != b).sum()))
    if M == 5000:
        with _opt("gemm_dyn", 1):
            out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            for _ in range(1100):
                ops.gemm_nt(A, W, bias=bias, out=out)
            torch.cuda.synchronize()
            assert torch.equal(out, ref[0])


def test_folded_target_forward_vs_the_emulating_oracle():
    """The D = 64 fixture model (weights and inputs produced by the REAL reference): target features h with the LayerNorms folded
    against the oracle that emulates the fold's storage points (oracle.EMU_TARGET_LN_FOLD: x_hat kept in fp32, bf16(W gamma), fp32
    bias b + W beta) -- 1e-3 like the unfused path against its emulation (tests/test_emu_parity_gpu.py), and within 2e-2 of the fp32 oracle."""
    import copy
    from oracle import vjepa_oracle as O
    from jepa_amd.engine.step import Trainer
    from tests.golden_util import HP, MICRO, load_micro, micro_weights, rel_l2, step_inputs
    from tests.test_step_gpu import build_micro_modules, load_into
    z = load_micro()
    enc_w, pred_w = micro_weights(z)
    enc, pred = build_micro_modules()
    load_into(enc, enc_w)
    load_into(pred, pred_w)
    tgt = copy.deepcopy(enc)
    for p in tgt.parameters():
        p.requires_grad = False
    enc.to(DEV), pred.to(DEV), tgt.to(DEV)
    tr = Trainer(enc, pred, tgt, loss_exp=HP["loss_exp"], reg_coeff=HP["reg_coeff"], betas=HP["betas"], eps=HP["eps"], device=DEV)
    tr.set_ln_fold(True)
    state = dict(enc=enc_w, pred=pred_w, tgt={k: v.clone() for k, v in enc_w.items()}, opt={})
    clips, me, mp = step_inputs(z, 0)
    o32, _ = O.step_grads(state, clips, me, mp, dict(MICRO), HP)
    old = O.EMU_TARGET_LN_FOLD
    O.EMU_TARGET_LN_FOLD = True
    try:
        oe, _ = O.step_grads(state, clips, me, mp, dict(MICRO), HP, emu=True)
    finally:
        O.EMU_TARGET_LN_FOLD = old
    cd, med, mpd = to_dev(clips, me, mp)
    h = tr.forward_target(cd, mpd)
    for i in range(len(mp)):
        eh, e32 = rel_l2(h[i].cpu(), oe["h"][i]), rel_l2(h[i].cpu(), o32["h"][i])
        print(f"[micro, folded target] mask {i}: h vs the fold-emulating oracle {eh:.2e} (fp32 oracle {e32:.2e})")
        assert eh < 1e-3 and e32 < 2e-2, (i, eh, e32)


@pytest.mark.parametrize("M,N,K", [(5000, 1288, 256), (10560, 3072, 1088)])
def test_non_temporal_operand_hint_is_bit_identical(ops, M, N, K):
    """Option gemm_nt (the LDS-DMA of one operand of the persistent GEMM carries the non-temporal cache hint): a cache policy, not
    arithmetic -- the same bits, for both choices of the operand and both families of tile orders."""
    g = torch.Generator(device=DEV).manual_seed(67)
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)

    def run_all():
        outs = [ops.gemm_nt(A, W, bias=bias), ops.gemm_nt(A, W, bias=bias, residual=res), ops.gemm_nt(A, W, bias=bias, epilogue=ops.EPI_GELU)]
        torch.cuda.synchronize()
        return outs
    ref = run_all()
    for nt in (1, 2):
        for raster in (260, 8):
            with _opt("gemm_nt", nt), _opt("gemm_raster", raster):
                for a, b in zip(ref, run_all()):
                    assert torch.equal(a, b), (nt, raster)


# ------------------------------------------------------------------------------------------ row operand of the epilogue requested up front
@pytest.mark.timeout(300)
@pytest.mark.parametrize("M,N,K", [(37632, 1024, 256), (5000, 1288, 256), (10560, 1024, 1088), (2304 + 40, 2600, 320), (58560, 384, 384)])
def test_epilogue_operand_preload_is_bit_identical(ops, M, N, K):
    """Option gemm_epi_pre = 1 / 2 / 3 (2 is the default): the residual (EPI_BF16) / the saved gelu' (EPI_DGELU, with and without fused column sums) of a wave
    tile is requested in one go before the epilogue's vmcnt(0) -- in the MFMA layout (1) or as full-line 16-byte loads re-laid-out
    through the staging area (2) -- instead of one 16-row block ahead.  Same values into the same arithmetic: the bits of the default,
    also on shifted edge tiles, with the dynamic tile hand-out and with the full grid; launches without a row operand keep their kernel."""
    g = torch.Generator(device=DEV).manual_seed(61)
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
    aux = (torch.rand(M, N, device=DEV, generator=g) * 1.2 - 0.1).to(torch.bfloat16)

    def run_all():
        outs = [ops.gemm_nt(A, W, bias=bias, residual=res), ops.gemm_nt(A, W, residual=res), ops.gemm_nt(A, W, bias=bias)]
        du, colpart = ops.gemm_dgelu_colsum(A, W, aux)
        outs.append(du)
        if colpart is not None:
            outs.append(colpart)
        outs.append(ops.gemm_nt(A, W, aux_in=aux, epilogue=ops.EPI_DGELU))
        torch.cuda.synchronize()
        return outs
    with _opt("gemm_epi_pre", 0):
        ref = run_all()
    for pre in (1, 2, 3):
        for persist, dyn in ((1, 0), (2, 0), (1, 1)):
            with _opt("gemm_epi_pre", pre), _opt("gemm_persist", persist), _opt("gemm_dyn", dyn):
                for rep in range(2):
                    got = run_all()
                    for i, (a, b) in enumerate(zip(ref, got)):
                        assert torch.equal(a, b), (pre, persist, dyn, rep, i, int((a != b).sum()))


# ------------------------------------------------------------------------------------------ pipelined epilogue passes
@pytest.mark.timeout(300)
@pytest.mark.parametrize("M,N,K", [(37632, 1152, 256), (5000, 1296, 256), (10560, 1536, 1088), (2304 + 40, 2592, 320), (58560, 384, 384)])
def test_pipelined_epilogue_is_bit_identical(ops, M, N, K):
    """Option gemm_epi_pre = 4 (default): the persistent kernel's epilogue passes are software-pipelined (the row-major read-back of pass ps is in
    flight while the arithmetic of pass ps + 1 runs; a row operand is parked and re-read for the next pass behind the issued reads).
    Every epilogue the persistent kernel has -- plain / bias / residual, the q-column scale, GELU with one and two outputs (both GELU
    forms), dGELU with and without the fused column sums, the folded LayerNorm -- must give the bits of the straight form (option 0),
    also on shifted edge tiles, with the full grid and with the dynamic tile hand-out."""
    g = torch.Generator(device=DEV).manual_seed(67)
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
    aux = (torch.rand(M, N, device=DEV, generator=g) * 1.2 - 0.1).to(torch.bfloat16)
    Wf32 = torch.randn(N, K, device=DEV, generator=g) * 0.05
    gamma = 1.0 + 0.3 * torch.randn(K, device=DEV, generator=g)
    beta = 0.2 * torch.randn(K, device=DEV, generator=g)
    rs = ops.ln_rowstats(A, 1e-6)
    Wf, cvec, bfold = ops.ln_fold_weights(Wf32, bias, gamma, beta)
    qs = 0.125 * 1.4426950408889634

    def run_all():
        outs = [ops.gemm_nt(A, W), ops.gemm_nt(A, W, bias=bias), ops.gemm_nt(A, W, bias=bias, residual=res), ops.gemm_nt(A, W, residual=res)]
        if N % 12 == 0:
            outs.append(ops.gemm_nt(A, W, bias=bias, epilogue=ops.EPI_QKV, alpha=qs))
            outs.append(ops.gemm_nt_lnfold(A, Wf, bfold, rs, cvec, epilogue=ops.EPI_QKV, alpha=qs))
        for poly in (1, 0):
            with _opt("gelu_poly", poly):
                outs.append(ops.gemm_nt(A, W, bias=bias, epilogue=ops.EPI_GELU))
                d = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
                outs.append(ops.gemm_nt(A, W, bias=bias, aux_out=d, epilogue=ops.EPI_GELU))
                outs.append(d)
                outs.append(ops.gemm_nt_lnfold(A, Wf, bfold, rs, cvec, epilogue=ops.EPI_GELU))
        outs.append(ops.gemm_nt_lnfold(A, Wf, bfold, rs, cvec))
        du, colpart = ops.gemm_dgelu_colsum(A, W, aux)
        outs.append(du)
        if colpart is not None:
            outs.append(colpart)
        outs.append(ops.gemm_nt(A, W, aux_in=aux, epilogue=ops.EPI_DGELU))
        outs.append(ops.gemm_nt(A, W, bias=bias, aux_in=aux, epilogue=ops.EPI_DGELU))
        torch.cuda.synchronize()
        return outs
    with _opt("gemm_epi_pre", 0):
        ref = run_all()
    for pre in (4,):
        for persist, dyn in ((1, 0), (2, 0), (1, 1)):
            with _opt("gemm_epi_pre", pre), _opt("gemm_persist", persist), _opt("gemm_dyn", dyn):
                for rep in range(2):
                    got = run_all()
                    assert len(got) == len(ref)
                    for i, (a, b) in enumerate(zip(ref, got)):
                        assert torch.equal(a, b), (pre, persist, dyn, rep, i, int((a != b).sum()))
