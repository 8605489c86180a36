"""Row kernels (csrc/rows.hip): mask gather / scatter (reference src/masks/utils.py:11-23) bit-exact, tubelet packing = Conv3d unfold
(src/models/utils/patch_embed.py:56), position add, transposes / column sums, predictor token assembly (src/models/predictor.py:194-221),
segment reductions, and the ds_read_b64_tr_b16 lane mapping the attention / TN-GEMM kernels rely on."""
import math
import pytest
import torch
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import rel_l2  # noqa: E402
from tests.step_util import TINY, TINY_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402
from tests.gpu_util import ATTN_SHAPES, bf, sdpa_ref  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def bf(x):
    return x.to(torch.bfloat16)


DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from jepa_amd.hip import ops as o
    return o


# ------------------------------------------------------------------------------------------ rows
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,N,K,D", [(2, 64, 20, 192), (3, 1568, 366, 1024), (1, 7, 7, 24), (4, 100, 1, 1536)])
def test_gather_scatter_bit_exact(ops, dtype, B, N, K, D):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, N, D, generator=g).to(dtype).to(DEV)
    idx = torch.stack([torch.randperm(N, generator=g)[:K].sort().values for _ in range(B)]).to(DEV)
    out = ops.gather_rows(x, idx)
    ref = torch.gather(x, 1, idx.unsqueeze(-1).repeat(1, 1, D))
    assert torch.equal(out.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                       ref.view(torch.int16 if dtype == torch.bfloat16 else torch.int32))
    # broadcast table (pos-embed) form
    tab = x[0:1].contiguous()
    out_b = ops.gather_rows(tab, idx)
    ref_b = torch.gather(tab.repeat(B, 1, 1), 1, idx.unsqueeze(-1).repeat(1, 1, D))
    assert torch.equal(out_b, ref_b)
    # scatter = backward of gather
    back = ops.scatter_rows(out, idx, N)
    ref_back = torch.zeros_like(x).scatter_(1, idx.unsqueeze(-1).repeat(1, 1, D), ref)
    assert torch.equal(back, ref_back)


def test_gather_empty(ops):
    x = torch.randn(2, 8, 16, device=DEV)
    idx = torch.zeros(2, 0, dtype=torch.int64, device=DEV)
    assert ops.gather_rows(x, idx).shape == (2, 0, 16)


@pytest.mark.parametrize("B,T,H,W,masked", [(2, 8, 64, 64, False), (2, 8, 64, 64, True), (2, 16, 224, 224, True)])
def test_tubelet_pack_matches_conv3d_unfold(ops, B, T, H, W, masked):
    g = torch.Generator().manual_seed(1)
    clips = torch.randn(B, 3, T, H, W, generator=g).to(DEV)
    tub, p = 2, 16
    N = (T // tub) * (H // p) * (W // p)
    # reference im2col with Conv3d ordering: token (t',h',w'), element (c,dt,dh,dw)
    u = clips.reshape(B, 3, T // tub, tub, H // p, p, W // p, p).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, N, -1)
    idx = None
    ref = u
    if masked:
        K = N // 3
        idx = torch.stack([torch.randperm(N, generator=g)[:K].sort().values for _ in range(B)]).to(DEV)
        ref = torch.gather(u, 1, idx.unsqueeze(-1).repeat(1, 1, u.shape[-1]))
    out = ops.tubelet_pack(clips, tub, p, idx)
    assert torch.equal(out.view(B, -1, u.shape[-1]), bf(ref))


def test_add_pos(ops):
    g = torch.Generator().manual_seed(2)
    B, N, K, D = 3, 64, 20, 192
    x = bf(torch.randn(B * K, D, generator=g)).to(DEV)
    pos = torch.randn(N, D, generator=g).to(DEV)
    idx = torch.stack([torch.randperm(N, generator=g)[:K].sort().values for _ in range(B)]).to(DEV)
    ref = bf(x.float() + pos[idx.reshape(-1)])
    out = ops.add_pos(x.clone(), pos, B, K, idx)
    assert torch.equal(out, ref)
    x2 = bf(torch.randn(B * N, D, generator=g)).to(DEV)
    ref2 = bf(x2.float() + pos.repeat(B, 1))
    assert torch.equal(ops.add_pos(x2.clone(), pos, B, N, None), ref2)


@pytest.mark.parametrize("M,N", [(100, 64), (473, 1024), (64, 3072), (1, 8)])
def test_transpose_and_colsum(ops, M, N):
    g = torch.Generator().manual_seed(3)
    x = bf(torch.randn(M, N, generator=g)).to(DEV)
    t = ops.transpose(x)
    Mp = ops.pad64(M)
    assert t.shape == (N, Mp)
    assert torch.equal(t[:, :M], x.t())
    assert torch.count_nonzero(t[:, M:]) == 0
    out = torch.full((N,), 3.0, device=DEV)
    ops.colsum(x, out, alpha=0.5, accumulate=True)
    ref = 0.5 * x.float().sum(0) + 3.0
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-4)


def test_colsum_row_window(ops):
    g = torch.Generator().manual_seed(4)
    B, Ke, Kp, D = 3, 5, 9, 96
    x = bf(torch.randn(B * (Ke + Kp), D, generator=g)).to(DEV)
    out = torch.zeros(D, device=DEV)
    ops.colsum(x, out, group=Ke + Kp, row_lo=Ke, row_hi=Ke + Kp)
    ref = x.float().view(B, Ke + Kp, D)[:, Ke:].sum((0, 1))
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------------------ predictor / loss / optimizer
def test_pred_assemble(ops):
    g = torch.Generator().manual_seed(11)
    B, N, Ke, Kp, D = 3, 64, 10, 30, 96
    e = bf(torch.randn(B * Ke, D, generator=g)).to(DEV)
    tok = torch.randn(D, generator=g).to(DEV)
    pos = torch.randn(N, D, generator=g).to(DEV)
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
    idx_e, idx_p = perm[:, :Ke].sort().values.to(DEV), perm[:, Ke:Ke + Kp].sort().values.to(DEV)
    out = ops.pred_assemble(e, tok, pos, idx_e, idx_p).view(B, Ke + Kp, D)
    ref_c = e.float().view(B, Ke, D) + pos[idx_e]
    ref_t = tok + pos[idx_p]
    assert torch.equal(out, bf(torch.cat([ref_c, ref_t], 1)))


def test_probe_tr16_dump(ops):
    """Record the ds_read_b64_tr_b16 lane mapping (not yet relied upon by any kernel)."""
    import ctypes
    import os
    from jepa_amd.hip.lib import load_library, check
    lib = load_library()
    res = {}
    for scale in (8, 16, 32):
        out = torch.zeros(256, dtype=torch.int32, device=DEV)
        check(lib.vj_probe_tr16(ctypes.c_void_p(out.data_ptr()), scale,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "probe")
        res[scale] = out.cpu().view(64, 4).tolist()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/probe_tr16.txt", "w") as f:
        for scale, rows in res.items():
            f.write(f"addr = base + {scale}*lane\n")
            for lane, r in enumerate(rows):
                f.write(f"  lane {lane:2d}: {r}\n")
    expect = [[(l & 15) + 16 * j + 64 * (l >> 4) for j in range(4)] for l in range(64)]
    print("tr16 mapping matches guide formula:", res[8] == expect)


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

