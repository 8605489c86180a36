"""Round-4 GPU tests: the seeded soft-max attention kernels (option attn_softmax) against the round-3 kernels and fp32
PyTorch, the re-base paths of the forward, bias-gradient column partials out of the producing kernels (attention backward,
fc2-dgrad epilogue), the one-launch segmented reduction, and the block chain with / without option bias_fuse.
Stated tolerances are next to each assertion; every kernel is reached through the C ABI."""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import rel_l2  # noqa: E402
from tests.step_util import TINY, TINY_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402
from tests.test_kernels_gpu import ATTN_SHAPES, bf, sdpa_ref  # noqa: E402

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
    """with _opt("name", value): ... restores the previous value."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        from jepa_amd.hip.lib import set_option
        self.old = set_option(self.name, self.value)

    def __exit__(self, *a):
        from jepa_amd.hip.lib import set_option
        set_option(self.name, self.old)


# ------------------------------------------------------------------------------------------ seeded soft-max attention
@pytest.mark.parametrize("B,S,H,hd", ATTN_SHAPES)
def test_seeded_softmax_attention_matches_round3_kernels_and_fp32(ops, B, S, H, hd):
    """Option attn_softmax = 1 (scale folded into the stationary operand with one more bf16 rounding, score accumulators
    seeded with -max / -lse, OR-of-exponent-bits re-base test, head_dim 24 row sums on the V pad column) against fp32 SDPA
    with the SAME bounds as the round-3 kernels (8e-3 forward, 1.5e-2 backward), and against the round-3 kernels themselves
    (<= 6e-3 forward / 1.2e-2 backward: two bf16 pipelines of the same arithmetic)."""
    g = torch.Generator().manual_seed(9)
    qkv = bf(torch.randn(B * S, 3 * H * hd, generator=g)).to(DEV)
    dout = bf(torch.randn(B * S, H * hd, generator=g)).to(DEV)
    scale = hd ** -0.5
    res = {}
    for sm in (0, 1):
        with _opt("attn_softmax", sm):
            o, lse = ops.attn_fwd(qkv, B, S, H, hd, scale)
            dqkv = ops.attn_bwd(qkv, o, dout, lse, B, S, H, hd, scale)
            torch.cuda.synchronize()
        res[sm] = (o.clone(), lse.clone(), dqkv.clone())
    x = qkv.float().requires_grad_(True)
    o_ref = sdpa_ref(x, B, S, H, hd)
    q, k, _ = x.detach().view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    lse_ref = torch.logsumexp((q @ k.transpose(-1, -2)) * scale, -1) / math.log(2.0)
    o_ref.backward(dout.float())
    gref = x.grad.view(B, S, 3, H, hd)
    for sm in (0, 1):
        o, lse, dqkv = res[sm]
        e = rel_l2(o, o_ref)
        assert e < 8e-3, ("fwd", sm, e)
        assert torch.allclose(lse, lse_ref, rtol=1e-3, atol=2e-2), (sm, float((lse - lse_ref).abs().max()))
        gout = dqkv.float().view(B, S, 3, H, hd)
        for i, name in enumerate(["dq", "dk", "dv"]):
            e = rel_l2(gout[:, :, i], gref[:, :, i])
            assert e < 1.5e-2, (name, sm, e)
    assert rel_l2(res[1][0], res[0][0].float()) < 6e-3
    assert rel_l2(res[1][2], res[0][2].float()) < 1.2e-2


@pytest.mark.parametrize("hd", [64, 24, 80])
def test_seeded_softmax_rebase_paths(ops, hd):
    """The forward keeps its base 2^5 above the largest score seen when the base was set and re-bases (exact maximum) only when
    a probability reaches 2.0.  Rows built to hit every path, against fp32 SDPA (max abs error 3e-2 as in the round-3
    forced-rescale test):
      row 1: scores rise by ~+8 (log2 units) at every key tile -> a re-base per tile;
      row 2: first tile far below zero (-60), later tiles around zero -> large upward re-base after a tiny first base;
      row 3: one key in the LAST tile 160 log2 units above everything -> exp2 overflows to +inf in the fast path (caught by
             the exponent test) and the re-based recomputation must be exact;
      row 4: all scores equal (the base never moves);  every other row random."""
    B, S, H = 1, 300, 1
    g = torch.Generator().manual_seed(10)
    t = torch.randn(B, S, 3, H, hd, generator=g)
    c = 1.0 / (hd ** -0.5 * math.log2(math.e))          # raw q.k product per log2 unit of score
    t[0, :, 1, 0, 0] = 0.0                              # key feature 0 is the handle: score += q0 * k0
    t[0, :, 0, 0, 0] = 0.0
    t[0, 1, 0, 0, 0] = 1.0
    t[0, :, 1, 0, 0] = (torch.arange(S) // 64).float() * 8.0 * c          # seen by query row 1 only (q0 = 1)
    t[0, 2, 0, 0] = 0.0
    t[0, 2, 0, 0, 1] = 1.0
    t[0, :, 1, 0, 1] = 0.0
    t[0, :64, 1, 0, 1] = -60.0 * c                                        # query row 2: first tile at -60
    t[0, 3, 0, 0] = 0.0
    t[0, 3, 0, 0, 2] = 4.0
    t[0, :, 1, 0, 2] = 0.0
    t[0, 290, 1, 0, 2] = 40.0 * c                                         # query row 3: key 290 at +160
    t[0, 4, 0, 0] = 0.0
    qkv = bf(t.reshape(B * S, -1)).to(DEV)
    ref = sdpa_ref(qkv, B, S, H, hd)
    # The folded scale costs one more bf16 rounding of the stationary operand, i.e. 2^-9 RELATIVE on every score: the ramp row
    # (scores up to 32 log2 units) carries up to 0.06 units of it = 4 % on a probability, the one-hot row (160) more, and the dK/dV
    # kernel (scale on K) and the forward (scale on Q) round differently.  Rows built on scores of that size are therefore
    # reproduced to a few per cent, not to bf16 precision; at |score| <= 10 the same term is <= 1.4 % and the random-input cases
    # above stay at 3e-3.  (The ordinary rows of THIS input see the same huge key features through their random q: their scores
    # are tens of log2 units too.)  Bounds: forward max abs error 1e-1 (folded scale) / 3e-2 (round-3 kernels); backward rel-L2
    # 4e-2 / 2e-2.
    for sm, bound in ((1, 1e-1), (0, 3e-2)):
        with _opt("attn_softmax", sm):
            o, lse = ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5)
        assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(lse).all())
        err = (o.float() - ref).abs()
        print(f"adversarial rows hd={hd} attn_softmax={sm}: forward max abs error {float(err.max()):.3e} "
              f"(rows 1-4: {[round(float(err[r].max()), 4) for r in (1, 2, 3, 4)]}, other rows {float(err[5:].max()):.3e})")
        assert float(err.max()) < bound, (sm, float(err.max()))
        if sm == 1:
            e_sm1 = float(err.max())
    # the same rows with the scale applied to q ONCE, before its only rounding (what the qkv GEMM epilogue does under option
    # attn_softmax = 2): the seeded kernels are then as exact as the round-3 ones -- forward 3e-2
    c = hd ** -0.5 * math.log2(math.e)
    tq = t.clone()
    tq[:, :, 0] = tq[:, :, 0] * c
    pre = bf(tq.reshape(B * S, -1)).to(DEV)
    xr = pre.float().view(B * S, 3, H * hd).clone()
    xr[:, 0] = xr[:, 0] / c
    ref_pre = sdpa_ref(xr.view(B * S, -1), B, S, H, hd)
    with _opt("attn_softmax", 1):
        o_p, _ = ops.attn_fwd(pre, B, S, H, hd, -(hd ** -0.5))
    e_p = float((o_p.float() - ref_pre).abs().max())
    print(f"adversarial rows hd={hd}, q pre-scaled before its rounding: forward max abs error {e_p:.3e}")
    assert e_p < 3e-2, e_p
    if hd == 24:   # the same rows with the row sums on the vector pipe: the pad-column sums must not be the less accurate ones
        with _opt("attn_softmax", 1), _opt("attn_psum", 0):
            o_v, _ = ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5)
        e_v = float((o_v.float() - ref).abs().max())
        print(f"adversarial rows hd=24 attn_softmax=1, row sums on the vector pipe: forward max abs error {e_v:.3e}")
        assert e_sm1 < 1.5 * e_v + 1e-2, (e_sm1, e_v)
    # and the backward consumes that lse (row 3: P is one-hot on key 290)
    dout = bf(torch.randn(B * S, H * hd, generator=g)).to(DEV)
    x = qkv.float().requires_grad_(True)
    sdpa_ref(x, B, S, H, hd).backward(dout.float())
    for sm, bound in ((1, 4e-2), (0, 2e-2)):
        with _opt("attn_softmax", sm):
            o2, lse2 = ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5)
            dqkv = ops.attn_bwd(qkv, o2, dout, lse2, B, S, H, hd, hd ** -0.5)
        assert bool(torch.isfinite(dqkv.float()).all())
        e = rel_l2(dqkv, x.grad)
        print(f"adversarial rows hd={hd} attn_softmax={sm}: backward rel-L2 {e:.2e}")
        assert e < bound, (sm, e)


# ------------------------------------------------------------------------------------------ soft-max scale applied by the qkv GEMM
LOG2E = 1.4426950408889634


@pytest.mark.parametrize("M,D,K", [(10560, 1024, 1024), (4000, 384, 384), (300, 192, 192), (2049, 1280, 1280)])
def test_qkv_gemm_epilogue_scales_the_q_columns_before_rounding(ops, M, D, K):
    """vj_gemm_bf16_nt epilogue 4: out[:, :N/3] = bf16((acc + bias) * alpha), the other two thirds bit-identical to epilogue 0;
    the q third within bf16 rounding of the fp32 product (rel-L2 4e-3).  Persistent kernel, one-tile kernel and the small generic
    kernel (M = 300) all take the column scale."""
    g = torch.Generator().manual_seed(81)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(3 * D, K, generator=g) * 0.05).to(DEV)
    bias = torch.randn(3 * D, generator=g).to(DEV)
    c = 0.125 * LOG2E
    plain = ops.gemm_nt(A, W, bias=bias)
    got = ops.gemm_nt(A, W, bias=bias, epilogue=ops.EPI_QKV, alpha=c)
    torch.cuda.synchronize()
    assert torch.equal(got[:, D:], plain[:, D:])
    ref_q = (A.float() @ W[:D].float().t() + bias[:D]) * c
    assert rel_l2(got[:, :D], ref_q) < 4e-3, rel_l2(got[:, :D], ref_q)
    # one rounding: the scaled q is NOT the re-rounded plain q (which is what scaling inside the attention kernels gives)
    twice = (plain[:, :D].float() * c).to(torch.bfloat16)
    assert rel_l2(got[:, :D], ref_q) <= rel_l2(twice, ref_q) + 1e-6


@pytest.mark.parametrize("B,S,H,hd", [(2, 366, 16, 64), (1, 1568, 4, 64), (2, 1113, 4, 24), (1, 65, 2, 24), (2, 200, 2, 80),
                                      (1, 129, 2, 128), (3, 52, 3, 32)])
def test_attention_with_prescaled_q(ops, B, S, H, hd):
    """scale < 0 = "the q part already carries |scale| * log2(e)" (what the chains do with option attn_softmax = 2): forward and
    backward against fp32 SDPA on the SAME operands (q_ref = q' / c), the round-3 bounds (8e-3 / 1.5e-2); dq is the gradient of
    the UNscaled q (what the qkv dgrad / wgrad consume)."""
    g = torch.Generator().manual_seed(91)
    qkv = bf(torch.randn(B * S, 3 * H * hd, generator=g))
    dout = bf(torch.randn(B * S, H * hd, generator=g)).to(DEV)
    scale = hd ** -0.5
    c = scale * LOG2E
    pre = qkv.clone().view(B * S, 3, H * hd)
    pre[:, 0] = (pre[:, 0].float() * c).to(torch.bfloat16)          # stored q' (here from the rounded q: any bf16 values will do)
    pre = pre.view(B * S, -1).to(DEV)
    xr = pre.float().view(B * S, 3, H * hd).clone()
    xr[:, 0] = xr[:, 0] / c                                          # the q the stored q' stands for
    xr = xr.view(B * S, -1).requires_grad_(True)
    o_ref = sdpa_ref(xr, B, S, H, hd)
    o_ref.backward(dout.float())
    for sm in (1, 0):                                                # the seeded kernels and the round-3 kernels both accept it
        with _opt("attn_softmax", sm):
            o, lse = ops.attn_fwd(pre, B, S, H, hd, -scale)
            dqkv = ops.attn_bwd(pre, o, dout, lse, B, S, H, hd, -scale)
            torch.cuda.synchronize()
        e = rel_l2(o, o_ref)
        assert e < 8e-3, ("fwd", sm, e)
        gref, gout = xr.grad.view(B, S, 3, H, hd), dqkv.float().view(B, S, 3, H, hd)
        errs = [rel_l2(gout[:, :, i], gref[:, :, i]) for i in range(3)]
        print(f"prescaled q, attn_softmax={sm}, B{B} S{S} H{H} hd{hd}: o {e:.2e} dq {errs[0]:.2e} dk {errs[1]:.2e} dv {errs[2]:.2e}")
        assert max(errs) < 1.5e-2, (sm, errs)


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


# ------------------------------------------------------------------------------------------ several segments, one launch
@pytest.mark.parametrize("H,hd,shapes", [(16, 64, [(3, 366), (3, 107)]), (16, 24, [(2, 1113), (2, 1208)]), (3, 32, [(2, 52), (0, 7), (2, 20)]),
                                         (2, 80, [(2, 200), (1, 63), (3, 129), (1, 16)]), (2, 128, [(1, 129), (2, 64)])])
@pytest.mark.parametrize("sm", [0, 1])
def test_attention_over_several_segments_in_one_launch(ops, H, hd, shapes, sm):
    """vj_attn_fwd_segs / vj_attn_bwd_segs (the masks of a batch concatenated along the rows, one launch for all of them) must give
    the bits of one vj_attn_fwd / vj_attn_bwd call per segment: o, lse2, dqkv and the column partials (segment after segment).
    Includes an empty segment and the 4-segment maximum."""
    g = torch.Generator().manual_seed(77)
    segs, r = [], 0
    for B, S in shapes:
        segs.append((r, B, S))
        r += B * S
    M = r
    qkv = bf(torch.randn(M, 3 * H * hd, generator=g)).to(DEV)
    dout = bf(torch.randn(M, H * hd, generator=g)).to(DEV)
    scale = hd ** -0.5
    with _opt("attn_softmax", sm):
        o, lse = ops.attn_fwd_segs(qkv, segs, H, hd, scale)
        dqkv, colq, colkv = ops.attn_bwd_segs(qkv, o, dout, lse, segs, H, hd, scale, colsum=True)
        dq2 = ops.attn_bwd_segs(qkv, o, dout, lse, segs, H, hd, scale)
        torch.cuda.synchronize()
        assert torch.equal(dq2, dqkv)
        oq = okv = 0
        for row0, B, S in segs:
            if B * S == 0:
                continue
            sl = slice(row0, row0 + B * S)
            o1, lse1 = ops.attn_fwd(qkv[sl], B, S, H, hd, scale)
            d1, cq1, ckv1 = ops.attn_bwd_colsum(qkv[sl], o1, dout[sl], lse1, B, S, H, hd, scale)
            torch.cuda.synchronize()
            assert torch.equal(o[sl], o1), (row0, "o")
            assert torch.equal(lse[H * row0:H * (row0 + B * S)].view(B, H, S), lse1), (row0, "lse")
            assert torch.equal(dqkv[sl], d1), (row0, "dqkv")
            assert torch.equal(colq[oq:oq + cq1.shape[0]], cq1) and torch.equal(colkv[okv:okv + ckv1.shape[0]], ckv1), (row0, "partials")
            oq, okv = oq + cq1.shape[0], okv + ckv1.shape[0]
        assert oq == colq.shape[0] and okv == colkv.shape[0]


# ------------------------------------------------------------------------------------------ bias-gradient column partials
@pytest.mark.parametrize("B,S,H,hd", [(2, 366, 16, 64), (3, 107, 16, 64), (2, 1113, 4, 24), (1, 65, 2, 24), (2, 200, 2, 80),
                                      (1, 129, 2, 128), (3, 52, 3, 32)])
@pytest.mark.parametrize("kt", [0, 1, 2])
def test_attention_backward_column_partials(ops, B, S, H, hd, kt):
    """vj_attn_bwd_colsum: dqkv bit-identical to vj_attn_bwd; the partial rows summed = column sums over the segment's tokens of
    the fp32 dQ | dK | dV (before their bf16 rounding): against the sums of the bf16 dqkv to 4e-3 rel-L2 (the rounding of
    B*S addends), and every partial row is written (NaN-poisoned buffers come back finite)."""
    g = torch.Generator().manual_seed(31)
    qkv = bf(torch.randn(B * S, 3 * H * hd, generator=g)).to(DEV)
    dout = bf(torch.randn(B * S, H * hd, generator=g)).to(DEV)
    scale = hd ** -0.5
    with _opt("attn_dkdv_kt", kt):
        o, lse = ops.attn_fwd(qkv, B, S, H, hd, scale)
        d0 = ops.attn_bwd(qkv, o, dout, lse, B, S, H, hd, scale).clone()
        d1, colq, colkv = ops.attn_bwd_colsum(qkv, o, dout, lse, B, S, H, hd, scale)
        torch.cuda.synchronize()
    assert torch.equal(d0, d1)
    assert bool(torch.isfinite(colq).all()) and bool(torch.isfinite(colkv).all())
    ref = d1.float().view(B * S, 3, H * hd).sum(0)                  # [3, H*hd]
    got = torch.cat([colq.sum(0), colkv.sum(0)]).view(3, H * hd)
    # the column sums of dK vanish in exact arithmetic (sum_k dS[q,k] = 0: a bias on k shifts every score of a query alike), so
    # both sides hold rounding noise there: errors are measured against the size of the dQ / dV sums
    size = float(torch.stack([ref[0], ref[2]]).norm()) / math.sqrt(2.0)
    for i, name in enumerate(["dq", "dk", "dv"]):
        e = float((got[i] - ref[i]).norm()) / size
        assert e < 4e-3, (name, e)
    assert float(got[1].norm()) <= float(ref[1].norm()) * 1.5 + 1e-3 * size   # the fp32 sums are at least as close to zero


@pytest.mark.parametrize("M,N,K", [(10560, 4096, 1024), (10000, 4096, 1024), (9999 // 8 * 8, 1536, 384), (300, 512, 256)])
def test_fc2_dgrad_epilogue_column_partials(ops, M, N, K):
    """vj_gemm_bf16_nt_dgelu_colsum: du bit-identical to the plain EPI_DGELU GEMM; where the persistent kernel takes the problem
    the partial rows sum to the column sums of the fp32 product (A W^T) * gelu' over ALL M rows exactly once -- M = 10000 has a
    SHIFTED last row tile (240 rows shared with its neighbour: counting them twice would be a 2.4 % error) -- to 1e-3 rel-L2
    against an fp32 PyTorch product; small problems fall back (no partials) and stay correct."""
    g = torch.Generator().manual_seed(41)
    A = bf(torch.randn(M, K, generator=g) / math.sqrt(K)).to(DEV)
    W = bf(torch.randn(N, K, generator=g)).to(DEV)
    aux = bf(torch.rand(M, N, generator=g) * 1.2 - 0.1).to(DEV)      # gelu' lives in [-0.13, 1.13]
    plain = ops.gemm_nt(A, W, aux_in=aux, epilogue=ops.EPI_DGELU)
    du, colpart = ops.gemm_dgelu_colsum(A, W, aux)
    torch.cuda.synchronize()
    assert torch.equal(plain, du)
    if M >= 4096:
        assert colpart is not None, "the persistent kernel should take this shape"
        assert bool(torch.isfinite(colpart).all())
        ref = ((A.float() @ W.float().t()) * aux.float()).sum(0)
        e = rel_l2(colpart.sum(0), ref)
        assert e < 1e-3, e
    else:
        assert colpart is None


def test_reduce_segments_matches_single_reductions(ops):
    """One launch, several independent reductions (ragged N, strided partial matrices, accumulate): bitwise equal to
    vj_reduce_partials run per segment, and close to torch's sums."""
    from jepa_amd.hip.lib import check, load_library
    g = torch.Generator().manual_seed(51)
    big = torch.randn(37, 3 * 192, generator=g).to(DEV)              # LayerNorm-style [nb][dgamma | dbeta | colsum]
    p2 = torch.randn(330, 96, generator=g).to(DEV)                   # N % 64 != 0
    p3 = torch.randn(5, 1024, generator=g).to(DEV)                   # fewer partial rows than partial lanes
    outs = [torch.randn(192, generator=g).to(DEV) for _ in range(3)] + [torch.randn(96, generator=g).to(DEV),
                                                                       torch.randn(1024, generator=g).to(DEV)]
    segs = [(big[:, 0:192], outs[0]), (big[:, 192:384], outs[1]), (big[:, 384:576], outs[2]), (p2, outs[3]), (p3, outs[4])]
    for alpha, acc in ((1.0, False), (0.25, True)):
        old = [o.clone() for o in outs]
        single = []
        lib = load_library()
        for (part, _), o0 in zip(segs, old):
            o = o0.clone()
            if part.stride(0) == part.shape[1]:
                check(lib.vj_reduce_partials(part.data_ptr(), o.data_ptr(), part.shape[0], part.shape[1], alpha, 1.0 if acc else 0.0,
                                             None), "vj_reduce_partials")
            else:
                c = part.contiguous()
                check(lib.vj_reduce_partials(c.data_ptr(), o.data_ptr(), c.shape[0], c.shape[1], alpha, 1.0 if acc else 0.0, None),
                      "vj_reduce_partials")
            single.append(o)
        ops.reduce_segments(segs, alpha=alpha, accumulate=acc)
        torch.cuda.synchronize()
        for (part, out), s1, o0 in zip(segs, single, old):
            assert torch.equal(out, s1)
            ref = alpha * part.double().sum(0) + (o0.double() if acc else 0.0)
            assert rel_l2(out.double().cpu(), ref.cpu()) < 1e-6


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


# ------------------------------------------------------------------------------------------ configs[3]: ViT-H in micro-batches of 24
@pytest.mark.timeout(1200)
def test_vith_micro_batches_of_24_reproduce_the_full_batch():
    """BASELINE configs[3] runs ViT-H/16 with 384 clips per GPU walked 24 at a time (bench.py --workload vith16).  The same
    structure at a size a single pass can still hold: B = 48 as one batch against two micro-batches of 24 (gradient
    accumulation through beta = 1 in every gradient writer, incl. the fused bias partials and the one reduction launch per block;
    losses accumulated on the device with the whole-batch normalisation).  lr = wd = 0, ema = 1: identical weights for both runs.
    Loss equal to 1e-6 relative, gradient arena rel-L2 <= 2e-5 (fp32 sums in a different order), every gradient finite."""
    from oracle import vjepa_oracle as O
    from tests.step_util import VITH, VITL_MASKS
    tr, _, _, _, _ = build_trainer(VITH, 2)
    gens = O.make_mask_gens(VITL_MASKS, VITH["crop"], VITH["frames"], VITH["patch"], VITH["tubelet"])
    clips, me, mp = draw_batch(gens, 48, VITH, 2024, 2025)
    cd, med, mpd = to_dev(clips, me, mp)
    out = {}
    for mb in (None, 24):
        tr.micro_batch = mb
        try:
            o = tr.train_step(cd, med, mpd, lr=0.0, wd=0.0, ema=1.0)
            torch.cuda.synchronize()
            out[mb] = (o.loss, tr.arena.G.clone(), o.skipped)
        finally:
            tr.micro_batch = None
    (l0, g0, sk0), (l1, g1, sk1) = out[None], out[24]
    assert not sk0 and not sk1 and bool(torch.isfinite(g0).all()) and bool(torch.isfinite(g1).all())
    assert abs(l1 - l0) <= 1e-6 * abs(l0), (l1, l0)
    r = float((g1.double() - g0.double()).norm() / g0.double().norm())
    print(f"ViT-H B=48: one batch vs 2 x 24: loss {l0:.6f} / {l1:.6f}, gradient arena rel-L2 {r:.2e}")
    assert r < 2e-5, r


# ------------------------------------------------------------------------------------------ GELU epilogue: exp2(polynomial) form
def _all_finite_bf16():
    bits = torch.arange(65536, dtype=torch.int32)
    x = (bits << 16).view(torch.float32)
    return x[torch.isfinite(x) & (x.abs() < 2.0 ** 126)]


@pytest.mark.parametrize("M", [512, 48])
def test_gelu_poly_epilogue(ops, M):
    """Option gelu_poly (default 1): the GELU epilogues of vj_gemm_bf16_nt over EVERY finite bf16 pre-activation (the GEMM only
    transports them: A = e_0 rows, W[:, 0] = the values, K = 256) against torch's float64 erf-GELU rounded to bf16 -- the reference's
    nn.GELU() (src/models/utils/modules.py:32).  M = 512: persistent 256 x 256 kernel (staged epilogue); M = 48: the small generic
    kernel (direct epilogue).  Bounds = what tests/test_gelu_poly.py finds for the same arithmetic on the CPU, plus the GPU's
    1-ulp v_exp_f32.  Option 0 (Abramowitz-Stegun 7.1.26) stays within its own, looser count; the two-output form (forward that
    saves gelu') returns the same GELU bit for bit and a derivative within bf16 rounding of autograd's."""
    v = _all_finite_bf16()
    N = (v.numel() + 255) // 256 * 256
    vals = torch.zeros(N)
    vals[: v.numel()] = v
    K = 256
    A = torch.zeros(M, K)
    A[:, 0] = 1.0
    W = torch.zeros(N, K)
    W[:, 0] = vals
    A, W = A.to(torch.bfloat16).to(DEV), W.to(torch.bfloat16).to(DEV)
    x64 = vals.double()
    exact = torch.nn.functional.gelu(x64)
    exact_b = exact.to(torch.bfloat16)
    inside = (vals > -5.0) & (vals.abs() > 2.0 ** -30)

    def check(out, max_diff, tail_abs):
        o = out.cpu()
        assert torch.equal(o, o[:1].expand_as(o))          # every row carries the same pre-activations
        o = o[0]
        assert torch.isfinite(o.float()).all()
        diff = inside & (o != exact_b)
        ulps = (o.view(torch.int16).int() - exact_b.view(torch.int16).int()).abs()
        n = int(diff.sum())
        assert n <= max_diff and int(ulps[diff].max() if n else 0) <= 1, (n, int(ulps[diff].max() if n else 0))
        assert float((o.double() - exact)[vals <= -5.0].abs().max()) < tail_abs
        return n

    with _opt("gelu_poly", 1):
        y1 = ops.gemm_nt(A, W, epilogue=ops.EPI_GELU)
        dg = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        y1b = ops.gemm_nt(A, W, aux_out=dg, epilogue=ops.EPI_GELU)
    with _opt("gelu_poly", 0):
        y0 = ops.gemm_nt(A, W, epilogue=ops.EPI_GELU)
    torch.cuda.synchronize()
    n1 = check(y1, 16, 2e-6)      # CPU restatement: 5 of 20.7 k inputs; the GPU's exp2 is 1 ulp, not correctly rounded
    n0 = check(y0, 48, 2e-6)      # CPU restatement of the A-S form: 22
    print(f"gelu epilogue M={M}: bf16 results differing from the correctly rounded erf-GELU: poly {n1}, A-S {n0}")
    assert torch.equal(y1b, y1)
    xg = x64.clone().requires_grad_(True)
    torch.nn.functional.gelu(xg).sum().backward()
    d = dg[0].cpu().double()
    assert float((d - xg.grad).abs().max()) < 5e-3        # half a bf16 ulp at 1.13 (4e-3) + the q error
    assert rel_l2(d, xg.grad) < 3e-3


# ------------------------------------------------------------------------------------------ tile order of the persistent GEMM
@pytest.mark.parametrize("M,N,K", [(8192 + 77, 2304, 512), (10560, 3072, 1088), (37632, 1024, 256), (58560, 384, 1536)])
def test_persistent_gemm_tile_orders_are_bit_identical(ops, M, N, K):
    """Option gemm_raster (group size, row- or column-grouped tile order of gemm8p.hip): a different ORDER of the same tiles -> the same bits,
    for the plain / residual / GELU epilogues and for the fc2-dgrad epilogue with its column partials (whose slot is the row tile)."""
    g = torch.Generator(device=DEV).manual_seed(41)
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
    aux = (torch.rand(M, N, device=DEV, generator=g) * 1.2 - 0.1).to(torch.bfloat16)

    def run_all():
        outs = [ops.gemm_nt(A, W, bias=bias), ops.gemm_nt(A, W, bias=bias, residual=res), ops.gemm_nt(A, W, bias=bias, epilogue=ops.EPI_GELU)]
        du, colpart = ops.gemm_dgelu_colsum(A, W, aux)
        outs.append(du)
        if colpart is not None:
            outs.append(colpart.sum(dim=0))
        torch.cuda.synchronize()
        return outs
    with _opt("gemm_raster", 0):
        ref = run_all()
    for raster in (4, 16, 2, 256 + 4, 256 + 2, 256 + 8, 256 + 6, 256 + 3, 511):
        with _opt("gemm_raster", raster):
            got = run_all()
        for i, (a, b) in enumerate(zip(ref, got)):
            if a.dtype == torch.float32:   # column sums: the same partial rows, summed here by torch (order-independent to 1e-6)
                assert rel_l2(b, a) < 1e-6, (raster, i)
            else:
                assert torch.equal(a, b), (raster, i, int((a != b).sum()))
