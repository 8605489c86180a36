"""bf16 MFMA GEMMs (csrc/gemm*.hip; every nn.Linear of the step: reference src/models/utils/modules.py:31-34,63,76) against fp32 PyTorch products,
the persistent kernel / the 4-wave kernel / every tile order / the pipelined epilogue bit-identical to their controls, the fused epilogues
(GELU + saved derivative over every finite bf16 input, q-column scale, fc1 bias partials, folded LayerNorm), weight gradients (transpose-free,
grouped, via transposes) and argument checks."""
import math
import pytest
import torch
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import rel_l2  # noqa: E402
from tests.step_util import TINY, TINY_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402
from tests.gpu_util import ATTN_SHAPES, bf, sdpa_ref  # noqa: E402,F401
import ctypes
from tests.step_util import TINY, TINY_MASKS, VITH, VITL, VITL_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from jepa_amd.hip import ops as _ops
    return _ops


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def bf(x):
    return x.to(torch.bfloat16)


# ------------------------------------------------------------------------------------------ gemm
GEMM_SHAPES = [
    (128, 128, 64), (473, 3072, 1024), (100, 1024, 4096), (37, 384, 1024), (1000, 1152, 384), (64, 576, 192),
    (33, 288, 96), (129, 132, 32), (4096, 4096, 1024), (256, 1024, 1536), (700, 520, 128), (2049, 1028, 320),
]


DEV = "cuda"


# ------------------------------------------------------------------------------------------------ persistent GEMM
def _gemm_case(M, N, K, epi, with_res, with_aux_out, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16) if with_res else None
    aux_in = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16) if epi == 2 else None
    return A, W, bias, res, aux_in


def _run_gemm(A, W, bias, res, aux_in, epi, with_aux_out):
    from jepa_amd.hip import ops
    M, N = A.shape[0], W.shape[0]
    out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    aux_out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16) if with_aux_out else None
    ops.gemm_nt(A, W, out=out, bias=None if epi == 2 else bias, residual=res, aux_in=aux_in, aux_out=aux_out, epilogue=epi)
    return out, aux_out


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


# ------------------------------------------------------------------------------------------ soft-max scale applied by the qkv GEMM
LOG2E = 1.4426950408889634


# ------------------------------------------------------------------------------------------ GELU epilogue: exp2(polynomial) form
def _all_finite_bf16():
    bits = torch.arange(65536, dtype=torch.int32)
    x = (bits << 16).view(torch.float32)
    return x[torch.isfinite(x) & (x.abs() < 2.0 ** 126)]


# ------------------------------------------------------------------------------------------ guard bands
BAND = 4096


PATTERN = 0xA5


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


@pytest.mark.parametrize("flags", [0, 0x20, 0x80, 0xC0, 0x100])   # auto | 256x256 | BK32 ring | 8-phase | 4-wave 2 WG/CU
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


def test_gemm_argument_errors(ops):
    from jepa_amd.hip.lib import HipKernelError
    A = torch.zeros(8, 40, dtype=torch.bfloat16, device=DEV)
    W = torch.zeros(8, 40, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(HipKernelError):
        ops.gemm_nt(A, W)  # K = 40 is not a multiple of 32


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


@pytest.mark.timeout(300)
@pytest.mark.parametrize("M,N,K", [
    (8192 + 77, 2304, 512),      # M edge (shifted last row tile), even K-tile count
    (8192, 2304 + 128, 320),     # N edge (shifted last column tile), ODD K-tile count: the ring half alternates per tile
    (37632, 1024, 256),          # the minimum K (HEAD and TAIL K-tiles back to back), 588 tiles
    (10560, 3072, 1088),         # a context-encoder row count of the step, 17 K-tiles
    (58560, 384, 1536),          # predictor fc2 shape: N = 1.5 tiles
])
def test_persistent_gemm_is_bit_identical_to_the_one_tile_kernel(M, N, K):
    """gemm8p.hip (one workgroup per CU walks its tiles, cross-tile prefetch, shifted edge tiles) must reproduce
    gemm8.hip bit for bit on every epilogue -- same K order, same epilogue arithmetic -- and both must sit within bf16
    rounding of an fp32 reference (rel-L2 <= 4e-3).  Three different operand draws per case (race screen)."""
    from jepa_amd.hip.lib import set_option
    cases = [(0, False, False), (0, True, False), (1, False, False), (1, False, True), (2, False, False)]
    for epi, with_res, with_aux in cases:
        for seed in range(3):
            ops_in = _gemm_case(M, N, K, epi, with_res, with_aux, 100 * epi + seed)
            old = set_option("gemm_persist", 0)
            try:
                ref, ref_aux = _run_gemm(*ops_in, epi, with_aux)
                # mode 1: trimmed grid (default), 2: one workgroup per CU -- both must give the one-tile kernel's bits
                for mode in (1, 2):
                    set_option("gemm_persist", mode)
                    got, got_aux = _run_gemm(*ops_in, epi, with_aux)
                    torch.cuda.synchronize()
                    assert not torch.isnan(got.float()).any(), (mode, epi, with_res, with_aux, seed, "unwritten output")
                    assert torch.equal(got, ref), (mode, epi, with_res, with_aux, seed, int((got != ref).sum()))
                    if with_aux:
                        assert torch.equal(got_aux, ref_aux), (mode, epi, seed, "aux")
            finally:
                set_option("gemm_persist", old)
        A, W, bias, res, aux_in = ops_in
        y = A.float() @ W.float().t()
        if epi != 2:
            y = y + bias
        if epi == 1:
            y = torch.nn.functional.gelu(y.to(torch.bfloat16).float())
        if epi == 2:
            y = y * aux_in.float()   # the saved GELU derivative
        if res is not None:
            y = y + res.float()
        assert rel_l2(got.float().cpu(), y.cpu()) < 4e-3, (epi, rel_l2(got.float().cpu(), y.cpu()))


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


# ------------------------------------------------------------------------------------------ pipelined epilogue passes
def _epilogue_suite(ops, M, N, K, seed=67):
    """A callable that runs EVERY epilogue the persistent kernel has on one (M, N, K) problem and returns the outputs."""
    g = torch.Generator(device=DEV).manual_seed(seed)
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
        outs.append(ops.gemm_nt(A, W, bias=bias, epilogue=ops.EPI_GELU))
        d = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        outs.append(ops.gemm_nt(A, W, bias=bias, aux_out=d, epilogue=ops.EPI_GELU))
        outs.append(d)
        outs.append(ops.gemm_nt_lnfold(A, Wf, bfold, rs, cvec, epilogue=ops.EPI_GELU))
        outs.append(ops.gemm_nt_lnfold(A, Wf, bfold, rs, cvec))
        du, colpart = ops.gemm_dgelu_colsum(A, W, aux)
        outs.append(du)
        outs.append(colpart)   # None when the column sums were not fused (one-tile kernel)
        outs.append(ops.gemm_nt(A, W, aux_in=aux, epilogue=ops.EPI_DGELU))
        outs.append(ops.gemm_nt(A, W, bias=bias, aux_in=aux, epilogue=ops.EPI_DGELU))
        torch.cuda.synchronize()
        return outs
    return run_all


@pytest.mark.timeout(300)
@pytest.mark.parametrize("M,N,K", [(37632, 1152, 256), (5000, 1296, 256), (10560, 1536, 1088), (2304 + 40, 2592, 320), (58560, 384, 384)])
def test_pipelined_epilogue_is_bit_identical(ops, M, N, K):
    """Option gemm_epi_pre = 4 (default): the persistent kernel's epilogue passes are software-pipelined (the row-major read-back of pass ps is in
    flight while the arithmetic of pass ps + 1 runs; a row operand is parked and re-read for the next pass behind the issued reads).
    Every epilogue the persistent kernel has -- plain / bias / residual, the q-column scale, GELU with one and two outputs, dGELU with
    and without the fused column sums, the folded LayerNorm -- must give the bits of the straight form (option 0, the A/B control), also
    on shifted edge tiles and with the full grid."""
    run_all = _epilogue_suite(ops, M, N, K)
    with _opt("gemm_epi_pre", 0):
        ref = run_all()
    for persist in (1, 2):
        with _opt("gemm_epi_pre", 4), _opt("gemm_persist", persist):
            for rep in range(2):
                got = run_all()
                assert len(got) == len(ref)
                for i, (a, b) in enumerate(zip(ref, got)):
                    assert (a is None) == (b is None), (persist, rep, i)
                    assert a is None or torch.equal(a, b), (persist, rep, i, int((a != b).sum()))


# ------------------------------------------------------------------------------------------ half tiles (N % 256 == 128)
@pytest.mark.timeout(300)
@pytest.mark.parametrize("M,N,K", [(58560, 384, 1536), (52800 + 8, 384, 384), (53760, 384, 1152), (3000, 384, 320), (256, 384, 256),
                                   (4000 + 24, 1152, 384)])
def test_half_tiles_are_bit_identical_to_full_tiles(ops, M, N, K):
    """Round 6: when N = 384 the shifted second column tile of the persistent kernel computes only the 128 columns it owns, on the
    early wave group (one wave per SIMD) while the late group idles through the barriers (N = 1152 keeps full tiles: control case).  Every epilogue must give the bits of the
    form that recomputes the overlap (option gemm_persist = 3, the A/B control) and of the one-tile kernel (gemm_persist = 0), with the
    trimmed and the full grid, on odd and even K-tile counts and with a shifted last ROW tile as well; outputs start as NaN, so a
    column nobody wrote would show."""
    run_all = _epilogue_suite(ops, M, N, K, seed=71)
    with _opt("gemm_persist", 3):
        ref = run_all()
    with _opt("gemm_persist", 0):
        one = run_all()
    for i, (a, b) in enumerate(zip(ref, one)):
        assert a is None or b is None or torch.equal(a, b), ("one-tile kernel", i, int((a != b).sum()))
    for persist in (1, 2):
        with _opt("gemm_persist", persist):
            for rep in range(2):
                got = None
                poison = [torch.full_like(t, float("nan")) for t in ref if t is not None]   # the allocator hands run_all() these blocks next
                del poison
                got = run_all()
                for i, (a, b) in enumerate(zip(ref, got)):
                    assert (a is None) == (b is None), (persist, rep, i)
                    if a is None:
                        continue
                    assert not torch.isnan(b.float()).any(), (persist, rep, i, "unwritten output")
                    assert torch.equal(a, b), (persist, rep, i, int((a != b).sum()))


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


@pytest.mark.parametrize("M", [512, 48])
def test_gelu_poly_epilogue(ops, M):
    """The GELU epilogues of vj_gemm_bf16_nt (Phi(-|x|) as exp2 of a degree-6 polynomial) over EVERY finite bf16 pre-activation (the GEMM only
    transports them: A = e_0 rows, W[:, 0] = the values, K = 256) against torch's float64 erf-GELU rounded to bf16 -- the reference's
    nn.GELU() (src/models/utils/modules.py:32).  M = 512: persistent 256 x 256 kernel (staged epilogue); M = 48: the small generic
    kernel (direct epilogue).  Bounds = what tests/test_gelu_poly.py finds for the same arithmetic on the CPU, plus the GPU's
    1-ulp v_exp_f32.  The two-output form (forward that saves gelu') returns the same GELU bit for bit and a derivative within bf16
    rounding of autograd's.  (The Abramowitz-Stegun form of rounds 1-3 differed in 22 inputs; it is gone since round 6.)"""
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

    y1 = ops.gemm_nt(A, W, epilogue=ops.EPI_GELU)
    dg = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    y1b = ops.gemm_nt(A, W, aux_out=dg, epilogue=ops.EPI_GELU)
    torch.cuda.synchronize()
    n1 = check(y1, 16, 2e-6)      # CPU restatement: 5 of 20.7 k inputs; the GPU's exp2 is 1 ulp, not correctly rounded
    print(f"gelu epilogue M={M}: bf16 results differing from the correctly rounded erf-GELU: {n1}")
    assert torch.equal(y1b, y1)
    xg = x64.clone().requires_grad_(True)
    torch.nn.functional.gelu(xg).sum().backward()
    d = dg[0].cpu().double()
    assert float((d - xg.grad).abs().max()) < 5e-3        # half a bf16 ulp at 1.13 (4e-3) + the q error
    assert rel_l2(d, xg.grad) < 3e-3


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


# ------------------------------------------------------------------------------------------------ grouped weight gradients
@pytest.mark.parametrize("T,D,Dh", [(1000, 256, 1024), (777, 384, 1536), (4160, 1024, 4096), (70000, 128, 264)])
def test_grouped_weight_gradients_match_the_single_launches(T, D, Dh):
    """vj_gemm_bf16_tn_grouped (qkv, proj, fc1, fc2 of a block in one launch) against four vj_gemm_bf16_tn_splitk launches
    and an fp64 reference, with alpha / beta accumulation.  The two kernels differ only in the split factor (fp32
    summation order): rel-L2 <= 2e-6 between them, <= 2e-3 to fp64 (bf16 inputs, K = T up to 70000)."""
    from jepa_amd.hip import ops
    g = torch.Generator(device=DEV).manual_seed(T)
    shapes = [(D, Dh), (Dh, D), (D, D), (3 * D, D)]            # (N1 = dY columns, N2 = X columns): fc2, fc1, proj, qkv
    probs, singles, olds = [], [], []
    for n1, n2 in shapes:
        dy = (torch.randn(T, n1, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
        x = torch.randn(T, n2, device=DEV, generator=g).to(torch.bfloat16)
        old = torch.randn(n1, n2, device=DEV, generator=g)
        olds.append(old)
        probs.append((dy, x, old.clone()))
        singles.append((dy, x, old.clone()))
    ops.gemm_wgrad_tn_grouped(probs, alpha=0.5, beta=1.0)
    for dy, x, out in singles:
        ops.gemm_wgrad_tn(dy, x, out, alpha=0.5, beta=1.0)
    torch.cuda.synchronize()
    for (dy, x, og), (_, _, os_), old in zip(probs, singles, olds):
        ref = 0.5 * (dy.double().t() @ x.double()) + old.double()
        assert rel_l2(og, os_) < 2e-6, rel_l2(og, os_)
        assert rel_l2(og, ref) < 2e-3, rel_l2(og, ref)
    # and a plain (beta = 0) run against fp64 directly
    outs = [torch.empty(n1, n2, device=DEV) for n1, n2 in shapes]
    ops.gemm_wgrad_tn_grouped([(p[0], p[1], o) for p, o in zip(probs, outs)], alpha=1.0, beta=0.0)
    for (dy, x, _), o in zip(probs, outs):
        ref = dy.double().t() @ x.double()
        assert rel_l2(o, ref.float()) < 2e-3, rel_l2(o, ref.float())
    again = [torch.empty_like(o) for o in outs]
    ops.gemm_wgrad_tn_grouped([(p[0], p[1], o) for p, o in zip(probs, again)], alpha=1.0, beta=0.0)
    for a, o in zip(again, outs):
        assert torch.equal(a, o)       # deterministic


def test_grouped_weight_gradients_argument_errors():
    from jepa_amd.hip import ops
    from jepa_amd.hip.lib import HipKernelError
    dy = torch.zeros(64, 16, device=DEV, dtype=torch.bfloat16)
    x = torch.zeros(64, 12, device=DEV, dtype=torch.bfloat16)      # 12 % 8 != 0
    with pytest.raises(HipKernelError):
        ops.gemm_wgrad_tn_grouped([(dy, x, torch.zeros(16, 12, device=DEV))])
    x8 = torch.zeros(64, 16, device=DEV, dtype=torch.bfloat16)
    five = [(dy, x8, torch.zeros(16, 16, device=DEV)) for _ in range(5)]
    with pytest.raises(HipKernelError):
        ops.gemm_wgrad_tn_grouped(five)

