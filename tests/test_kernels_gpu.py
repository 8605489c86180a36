"""Kernel-level parity: every HIP kernel behind the C ABI vs a plain PyTorch fp32 reference of the same op.

Tolerances: gather/scatter/pack index work is bit-exact; bf16-output kernels are compared after rounding the fp32
reference to bf16 with a few-ulp allowance (bf16 eps = 2^-8); fp32-accumulating reductions to 1e-5..1e-3 relative.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from jepa_amd.hip import ops as _ops
    return _ops


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def bf(x):
    return x.to(torch.bfloat16)


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


# ------------------------------------------------------------------------------------------ layernorm
@pytest.mark.parametrize("rows,D", [(40, 192), (473, 1024), (1000, 384), (7, 1280), (5, 96)])
def test_layernorm_fwd_bwd(ops, rows, D):
    g = torch.Generator().manual_seed(5)
    x = bf(torch.randn(rows, D, generator=g) * 2 + 0.3).to(DEV)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    dy = bf(torch.randn(rows, D, generator=g)).to(DEV)
    dres = bf(torch.randn(rows, D, generator=g)).to(DEV)
    eps = 1e-6
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps)
    xr = x.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, eps)
    assert rel_l2(y, yr) < 4e-3, rel_l2(y, yr)
    assert torch.allclose(mean, x.float().mean(-1), rtol=1e-4, atol=1e-5)
    yr.backward(dy.float())
    dgamma = torch.zeros(D, device=DEV)
    dbeta = torch.zeros(D, device=DEV)
    dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=dres)
    assert rel_l2(dx, xr.grad + dres.float()) < 6e-3, rel_l2(dx, xr.grad + dres.float())
    assert rel_l2(dgamma, gr.grad) < 1e-4, rel_l2(dgamma, gr.grad)
    assert rel_l2(dbeta, br.grad) < 1e-4, rel_l2(dbeta, br.grad)
    # accumulate + alpha
    dx2 = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, alpha=2.0, accumulate=True)
    assert rel_l2(dgamma, 3 * gr.grad) < 1e-4
    assert rel_l2(dx2, xr.grad) < 6e-3


# ------------------------------------------------------------------------------------------ gemm
GEMM_SHAPES = [
    (128, 128, 64), (473, 3072, 1024), (100, 1024, 4096), (37, 384, 1024), (1000, 1152, 384), (64, 576, 192),
    (33, 288, 96), (129, 132, 32), (4096, 4096, 1024), (256, 1024, 1536), (700, 520, 128), (2049, 1028, 320),
]


@pytest.mark.parametrize("flags", [0, 1, 0x20, 0x80, 0xC0, 0x100])   # auto | reg-staged | 256x256 | BK32 ring | 8-phase | 4-wave 2 WG/CU
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_plain_bias_residual(ops, flags, M, N, K):
    g = torch.Generator().manual_seed(6)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = bf(torch.randn(M, N, generator=g)).to(DEV)
    ref = A.float() @ W.float().t()
    out = ops.gemm_nt(A, W, flags=flags)
    assert rel_l2(out, ref) < 4e-3, rel_l2(out, ref)
    out = ops.gemm_nt(A, W, bias=bias, residual=res, flags=flags)
    ref2 = ref + bias + res.float()
    assert rel_l2(out, ref2) < 4e-3, rel_l2(out, ref2)
    # every pipeline / tile shape accumulates each output in the same k order -> identical bits
    if flags != 0:
        assert torch.equal(out, ops.gemm_nt(A, W, bias=bias, residual=res, flags=0))


@pytest.mark.parametrize("M,N,K", [(473, 4096, 1024), (100, 384, 96), (130, 1536, 384)])
def test_gemm_gelu_epilogues(ops, M, N, K):
    g = torch.Generator().manual_seed(7)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    u_ref = (A.float() @ W.float().t() + bias).to(torch.bfloat16).float()   # the epilogue rounds the pre-activation to bf16
    dg = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    gout = ops.gemm_nt(A, W, bias=bias, aux_out=dg, epilogue=ops.EPI_GELU)
    g_ref = torch.nn.functional.gelu(u_ref)  # exact erf GELU of the bf16 pre-activation
    assert rel_l2(gout, g_ref) < 4e-3, rel_l2(gout, g_ref)
    assert torch.equal(gout, ops.gemm_nt(A, W, bias=bias, epilogue=ops.EPI_GELU))   # saving the derivative changes nothing
    # aux_out = gelu'(u) (bf16): the derivative is saved instead of the pre-activation
    uu = u_ref.clone().requires_grad_(True)
    torch.nn.functional.gelu(uu).sum().backward()
    assert rel_l2(dg, uu.grad) < 4e-3, rel_l2(dg, uu.grad)
    assert (dg.float() - uu.grad).abs().max() < 2e-2   # half a bf16 ulp at 1.13 plus a one-ulp rounding flip of u itself
    # dgelu epilogue: out = (A W^T) * saved derivative
    d = ops.gemm_nt(A, W, aux_in=dg, epilogue=ops.EPI_DGELU)
    d_ref = (A.float() @ W.float().t()) * uu.grad
    assert rel_l2(d, d_ref) < 5e-3, rel_l2(d, d_ref)


@pytest.mark.parametrize("T,N1,N2", [(473, 1024, 1024), (11392, 3072, 1024), (1000, 520, 776), (66, 256, 256),
                                     (64, 8, 264), (4099, 1024, 4096)])
def test_gemm_wgrad_tn_without_transposes(ops, T, N1, N2):
    """dW[N1,N2] = alpha * dY[T,N1]^T X[T,N2] + beta * dW straight from the row-major operands (transpose reads in LDS,
    zero rows for the last partial 64-token tile, split-K over the tokens) vs fp32 torch and vs the transpose route."""
    g = torch.Generator().manual_seed(31)
    dY = bf(torch.randn(T, N1, generator=g)).to(DEV)
    X = bf(torch.randn(T, N2, generator=g)).to(DEV)
    out = torch.full((N1, N2), 1.0, device=DEV)
    ops.gemm_wgrad_tn(dY, X, out, alpha=0.5, beta=2.0)
    ref = 0.5 * (dY.float().t() @ X.float()) + 2.0
    assert rel_l2(out, ref) < 1e-5, rel_l2(out, ref)
    out2 = torch.empty((N1, N2), device=DEV)
    ops.gemm_wgrad_tn(dY, X, out2, alpha=0.25)
    via_t = torch.empty((N1, N2), device=DEV)
    ops.gemm_wgrad(ops.transpose(dY), ops.transpose(X), via_t, alpha=0.25)
    assert rel_l2(out2, via_t) < 2e-6, rel_l2(out2, via_t)


@pytest.mark.parametrize("M,N,K", [(1024, 384, 473), (3072, 1024, 1000), (96, 288, 66)])
def test_gemm_wgrad_fp32_via_transposes(ops, M, N, K):
    """dW[M=N_out, N=K_in] = dY^T X with K = tokens (padded to 64 by the transpose kernel)."""
    g = torch.Generator().manual_seed(8)
    tokens = K
    dY = bf(torch.randn(tokens, M, generator=g)).to(DEV)
    X = bf(torch.randn(tokens, N, generator=g)).to(DEV)
    dYt, Xt = ops.transpose(dY), ops.transpose(X)
    out = torch.full((M, N), 1.0, device=DEV)
    ops.gemm_nt(dYt, Xt, out=out, epilogue=ops.EPI_F32, alpha=0.5, beta=2.0)
    ref = 0.5 * (dY.float().t() @ X.float()) + 2.0
    assert rel_l2(out, ref) < 1e-5, rel_l2(out, ref)


def test_gemm_argument_errors(ops):
    from jepa_amd.hip.lib import HipKernelError
    A = torch.zeros(8, 40, dtype=torch.bfloat16, device=DEV)
    W = torch.zeros(8, 40, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(HipKernelError):
        ops.gemm_nt(A, W)  # K = 40 is not a multiple of 32


# ------------------------------------------------------------------------------------------ attention
def sdpa_ref(qkv, B, S, H, hd):
    q, k, v = qkv.float().view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    return o.transpose(1, 2).reshape(B * S, H * hd)


ATTN_SHAPES = [(2, 20, 3, 64), (2, 64, 3, 32), (1, 107, 16, 64), (2, 366, 16, 64), (1, 1568, 4, 64),
               (2, 300, 16, 24), (1, 1113, 4, 24), (1, 200, 2, 80), (3, 52, 3, 32), (1, 129, 2, 128),
               # ViT-H head_dim 80 on the native 96-wide class (3 k-steps, 5 output tiles): ragged, multi-tile, and the
               # full 384^2 x 16 frame sequence of BASELINE configs[4] (8 x 24 x 24 = 4608 tokens, 16 heads)
               (2, 63, 16, 80), (1, 1568, 2, 80), (1, 4608, 16, 80), (2, 577, 3, 72),
               # head_dim 24 / 32 on the re-swizzled 64-byte-row images: tile boundaries +-1
               (1, 64, 2, 24), (1, 65, 2, 24), (2, 127, 2, 32), (1, 1208, 16, 24)]


@pytest.mark.parametrize("B,S,H,hd", ATTN_SHAPES)
def test_attention_fwd_bwd(ops, B, S, H, hd):
    g = torch.Generator().manual_seed(9)
    qkv = bf(torch.randn(B * S, 3 * H * hd, generator=g)).to(DEV)
    dout = bf(torch.randn(B * S, H * hd, generator=g)).to(DEV)
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(qkv, B, S, H, hd, scale)
    x = qkv.float().requires_grad_(True)
    o_ref = sdpa_ref(x, B, S, H, hd)
    assert rel_l2(o, o_ref) < 8e-3, ("fwd", rel_l2(o, o_ref))
    # lse2 is log2-domain: check against explicit scores
    q, k, _ = x.detach().view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    lse_ref = torch.logsumexp((q @ k.transpose(-1, -2)) * scale, -1) / math.log(2.0)
    assert torch.allclose(lse, lse_ref, rtol=1e-3, atol=2e-2), (lse - lse_ref).abs().max()
    o_ref.backward(dout.float())
    dqkv = ops.attn_bwd(qkv, o, dout, lse, B, S, H, hd, scale)
    gref = x.grad.view(B, S, 3, H, hd)
    gout = dqkv.float().view(B, S, 3, H, hd)
    for i, name in enumerate(["dq", "dk", "dv"]):
        e = rel_l2(gout[:, :, i], gref[:, :, i])
        assert e < 1.5e-2, (name, e)


@pytest.mark.parametrize("B,S,H,hd", ATTN_SHAPES)
def test_attention_dkdv_two_key_tiles_per_wave_is_bit_identical(ops, B, S, H, hd):
    """attn_bwd_dkdv_kernel<HDP, 2> (32 keys per wave: every Q / dO fragment and transposed read serves two key tiles) accumulates each
    dK / dV element over the query tiles in the same order as the 16-keys-per-wave form: dqkv must be equal bit for bit.  Only the
    head-dim classes that instantiate both forms (hd <= 64) are compared."""
    if hd > 64:
        pytest.skip("one form only for this head-dim class")
    from jepa_amd.hip.lib import set_option
    g = torch.Generator().manual_seed(21)
    qkv = bf(torch.randn(B * S, 3 * H * hd, generator=g)).to(DEV)
    dout = bf(torch.randn(B * S, H * hd, generator=g)).to(DEV)
    o, lse = ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5)
    old = set_option("attn_dkdv_kt", 1)
    try:
        d1 = ops.attn_bwd(qkv, o, dout, lse, B, S, H, hd, hd ** -0.5).clone()
        set_option("attn_dkdv_kt", 2)
        d2 = ops.attn_bwd(qkv, o, dout, lse, B, S, H, hd, hd ** -0.5).clone()
        torch.cuda.synchronize()
    finally:
        set_option("attn_dkdv_kt", old)
    assert torch.equal(d1, d2), int((d1 != d2).sum())


def test_attention_forced_rescale(ops):
    """One key row spiked against one query row at a late tile: the online-softmax rescale path must be exact."""
    B, S, H, hd = 1, 300, 1, 64
    g = torch.Generator().manual_seed(10)
    t = torch.randn(B, S, 3, H, hd, generator=g)
    t[0, 5, 0, 0] *= 0  # q row 5
    t[0, 5, 0, 0, 0] = 30.0
    t[0, 250, 1, 0, 0] = 30.0  # k row 250 (tile 3) dominates q row 5
    qkv = bf(t.reshape(B * S, -1)).to(DEV)
    o, _ = ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5)
    ref = sdpa_ref(qkv, B, S, H, hd)
    assert (o.float() - ref).abs().max() < 3e-2


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


def test_target_rows_and_loss(ops):
    g = torch.Generator().manual_seed(12)
    B, N, K, D = 2, 64, 32, 192
    x = bf(torch.randn(B * N, D, generator=g) * 3).to(DEV)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    idx = torch.stack([torch.randperm(N, generator=g)[:K].sort().values for _ in range(B)]).to(DEV)
    h = ops.target_rows(x, gamma, beta, idx, B, N, 1e-6)
    F = torch.nn.functional
    hr = F.layer_norm(F.layer_norm(x.float().view(B, N, D), (D,), gamma, beta, 1e-6), (D,))
    hr = torch.gather(hr, 1, idx.unsqueeze(-1).repeat(1, 1, D))
    assert torch.allclose(h, hr, rtol=1e-4, atol=1e-4), (h - hr).abs().max()
    z = bf(torch.randn(B * K, D, generator=g)).to(DEV)
    loss = torch.zeros(1, device=DEV)
    dz = torch.empty_like(z)
    numel = z.numel()
    ops.latent_loss(z, h, loss, p=1.0, out_scale=1.0 / numel, dz=dz, gscale=0.25)
    ref = (z.float().view(B, K, D) - hr).abs().mean()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item()) + 1e-7
    assert torch.equal(dz.float().view(B, K, D), 0.25 * torch.sign(z.float().view(B, K, D) - h))
    loss2 = torch.ones(1, device=DEV)
    ops.latent_loss(z, h, loss2, p=2.0, out_scale=1.0 / numel, accumulate=True)
    ref2 = 1.0 + ((z.float().view(B, K, D) - hr).abs() ** 2).mean() / 2
    assert abs(loss2.item() - ref2.item()) < 1e-4 * ref2.item()
    # variance regulariser (reg_fn)
    pstd = torch.zeros(B, D, device=DEV)
    ops.token_pstd(z, pstd, B, K, D, False)
    ref_p = torch.sqrt(z.float().view(B, K, D).var(dim=1) + 1e-4)
    assert torch.allclose(pstd, ref_p, rtol=1e-4, atol=1e-5)
    out = torch.zeros(1, device=DEV)
    ops.reg_finish(pstd, 1, out)
    assert abs(out.item() - torch.relu(1 - ref_p).mean().item()) < 1e-5


def test_adamw_ema_matches_torch(ops):
    g = torch.Generator().manual_seed(13)
    n = 4096 + 64
    p0 = torch.randn(n, generator=g)
    tgt0 = p0.clone()
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    p = p0.clone().to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    tgt = tgt0.clone().to(DEV)
    pb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    tb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    tr = tgt0.clone()
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        pr.grad = grad.clone()
        opt.step()
        tr.mul_(0.998).add_((1 - 0.998) * pr.detach())
        ops.adamw_ema(p, grad.to(DEV), m, v, pb, tgt, tb, 1e-3, 0.05, 0.9, 0.999, 1e-8, step, 1.0, 0.998)
    assert torch.allclose(p.cpu(), pr.detach(), rtol=1e-5, atol=1e-6), (p.cpu() - pr.detach()).abs().max()
    assert torch.allclose(tgt.cpu(), tr, rtol=1e-5, atol=1e-6)
    assert torch.equal(pb, bf(p))
    assert torch.equal(tb, bf(tgt))
    out2 = torch.zeros(2, device=DEV)
    ops.sqnorm(p, out2)
    assert abs(out2[0].item() - (p.double() ** 2).sum().item()) < 1e-4 * out2[0].item()
    assert out2[1].item() == 0
    p[3] = float("nan")
    ops.sqnorm(p, out2)
    assert out2[1].item() == 1


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


@pytest.mark.parametrize("M,N,K", [(1024, 1024, 11392), (384, 384, 27900), (3072, 1024, 473), (96, 288, 66)])
def test_gemm_wgrad_splitk_and_fused_colsum(ops, M, N, K):
    """Split-K wgrad (deterministic slice reduction) + bias gradient fused into the dY transpose."""
    g = torch.Generator().manual_seed(14)
    dY = bf(torch.randn(K, M, generator=g)).to(DEV)
    X = bf(torch.randn(K, N, generator=g)).to(DEV)
    db = torch.full((M,), 2.0, device=DEV)
    dYt = ops.transpose_colsum(dY, db, alpha=0.5, accumulate=True)
    assert torch.equal(dYt[:, :K], dY.t()) and torch.count_nonzero(dYt[:, K:]) == 0
    assert torch.allclose(db, 0.5 * dY.float().sum(0) + 2.0, rtol=1e-5, atol=2e-3)
    Xt = ops.transpose(X)
    out = torch.full((M, N), 1.0, device=DEV)
    ops.gemm_wgrad(dYt, Xt, out, alpha=0.25, beta=3.0)
    ref = 0.25 * (dY.float().t() @ X.float()) + 3.0
    assert rel_l2(out, ref) < 1e-5, rel_l2(out, ref)
    out2 = torch.full((M, N), 1.0, device=DEV)
    ops.gemm_wgrad(dYt, Xt, out2, alpha=0.25, beta=3.0)
    assert torch.equal(out, out2)  # deterministic


# ------------------------------------------------------------------------------------------ 4-wave GEMM (gemm4w.hip)
@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (1000, 384, 384), (2311, 1152, 384), (4099, 1024, 1024),
                                   (256, 128, 64), (300, 200, 192), (37632, 1024, 1024)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_4wave_two_workgroups_per_cu_matches_8phase_bitwise(ops, M, N, K, epi):
    """The 256x128 / 4-wave / two-workgroups-per-CU kernel accumulates every output element over the same K-tile and
    k-step order as the 256x256 8-phase kernel and shares its epilogues: results must be BIT-identical (any DMA / LDS
    race of the new schedule shows up as a mismatch), for every epilogue, interior and edge tiles, K from one to 16
    K-tiles; repeated to catch timing-dependent races."""
    g = torch.Generator().manual_seed(77)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = bf(torch.randn(M, N, generator=g)).to(DEV)
    aux = bf(torch.randn(M, N, generator=g)).to(DEV)

    def run(flags):
        if epi == 0:
            return ops.gemm_nt(A, W, bias=bias, residual=res, flags=flags), None
        if epi == 1:
            u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            return ops.gemm_nt(A, W, bias=bias, aux_out=u, epilogue=ops.EPI_GELU, flags=flags), u
        if epi == 2:
            return ops.gemm_nt(A, W, aux_in=aux, epilogue=ops.EPI_DGELU, flags=flags), None
        out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        lib = ops.load_library()
        ops.check(lib.vj_gemm_bf16_nt(A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K, None, None, 0, None,
                                      None, 0, 3, 0.5, 0.0, flags, torch.cuda.current_stream().cuda_stream), "gemm f32")
        return out, None
    ref, ref_u = run(0xC0)         # 8-phase 256x256
    for _ in range(3):
        out, u = run(0x100)
        assert torch.equal(out, ref), float((out.float() - ref.float()).abs().max())
        if ref_u is not None:
            assert torch.equal(u, ref_u)
    if epi == 0:   # and against fp32 torch, like every other GEMM test
        r = A.float() @ W.float().t() + bias + res.float()
        assert rel_l2(out, r) < 4e-3
