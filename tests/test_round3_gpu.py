"""Round-3 GPU tests: input-edge staging safety, run-time options, the data-parallel reducer at one rank, arena-wide
gradient parity at the benched size, and the kernels added in round 3 (each checked against the kernel it replaces
and against an fp32 PyTorch reference).  Stated tolerances are next to each assertion."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import rel_l2  # noqa: E402
from tests.step_util import TINY, TINY_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gens(masks=TINY_MASKS, m=TINY):
    from oracle import vjepa_oracle as O
    return O.make_mask_gens(masks, m["crop"], m["frames"], m["patch"], m["tubelet"])


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
                # mode 1: trimmed grid (default), 2: one workgroup per CU; sched 8: four {load, compute} section pairs per K-tile,
                # 4 (round 4): two pairs of 32 MFMAs -- every combination must give the one-tile kernel's bits
                for mode, sched in ((1, 8), (2, 8), (1, 4), (2, 4)):
                    set_option("gemm_persist", mode)
                    old_s = set_option("gemm_sched", sched)
                    try:
                        got, got_aux = _run_gemm(*ops_in, epi, with_aux)
                        torch.cuda.synchronize()
                    finally:
                        set_option("gemm_sched", old_s)
                    assert not torch.isnan(got.float()).any(), (mode, sched, epi, with_res, with_aux, seed, "unwritten output")
                    assert torch.equal(got, ref), (mode, sched, epi, with_res, with_aux, seed, int((got != ref).sum()))
                    if with_aux:
                        assert torch.equal(got_aux, ref_aux), (mode, sched, epi, seed, "aux")
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


# ------------------------------------------------------------------------------------------------ LayerNorm backward + column sums
@pytest.mark.parametrize("rows,D", [(10560, 1024), (5533, 384), (777, 1280), (40, 192)])
def test_layernorm_bwd_column_sums_of_dx(rows, D):
    """vj_layernorm_bwd_colsum: dx / dgamma / dbeta equal to the plain backward (up to fp contraction in a separately
    compiled variant); dxsum = alpha * column sums of dx
    (accumulated in fp32 BEFORE the bf16 rounding of dx) within 2e-3 relative of the fp32 sum of the rounded dx, and
    accumulating (beta = 1) adds to the previous value."""
    from jepa_amd.hip import ops
    g = torch.Generator(device=DEV).manual_seed(rows + D)
    x = torch.randn(rows, D, device=DEV, generator=g).to(torch.bfloat16)
    dy = torch.randn(rows, D, device=DEV, generator=g).to(torch.bfloat16)
    dres = torch.randn(rows, D, device=DEV, generator=g).to(torch.bfloat16)
    gamma = torch.randn(D, device=DEV, generator=g)
    beta = torch.randn(D, device=DEV, generator=g)
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    outs = []
    for with_cs in (False, True):
        dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        cs = torch.full((D,), 3.0, device=DEV) if with_cs else None
        dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dres=dres, alpha=0.5, dxsum=cs)
        if with_cs:
            first = cs.clone()
            ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dres=dres, alpha=0.5, accumulate=True, dxsum=cs)
            assert torch.allclose(cs, 2 * first, rtol=1e-6, atol=1e-6)
            cs = first
        outs.append((dx, dg.clone(), db.clone(), cs))
    # the two template variants are compiled separately under -ffp-contract=fast: dx may differ by one bf16 ulp in a few
    # elements, never more (C chain and Python chain call the same variant for the same tensor: their bit-identity holds)
    da, dbb = outs[0][0].float(), outs[1][0].float()
    frac = float((da != dbb).float().mean())
    print(f"layernorm_bwd colsum variant vs plain, rows={rows} D={D}: {100 * frac:.4f} % of dx elements differ, rel-L2 "
          f"{float((da - dbb).norm() / da.norm()):.2e}")
    assert frac < 2e-3 and float((da - dbb).norm() / da.norm()) < 2e-4
    ref = 0.5 * outs[1][0].float().sum(0)
    err = float((outs[1][3] - ref).norm() / ref.norm())
    assert err < 2e-3, err
    # the first call of the with_cs arm ran with beta = 0: dgamma/dbeta equal the plain call's halves after the accumulate
    assert torch.allclose(outs[1][1], 2 * outs[0][1], rtol=1e-6, atol=1e-5)
    assert torch.allclose(outs[1][2], 2 * outs[0][2], rtol=1e-6, atol=1e-5)



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

# ------------------------------------------------------------------------------------------------ data-parallel reducer at one rank


def _reducer_worker(q):
    """Own process: torch.distributed must be initialised (and destroyed) exactly once per process."""
    try:
        import socket
        import torch.distributed as dist
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        res = {}
        for mode in ("none", "torch"):
            os.environ["VJ_FORCE_DP"] = "0" if mode == "none" else "1"
            tr, _, _, _, _ = build_trainer(TINY, 2, perturb_small=True)
            assert tr.reducer.enabled == (mode != "none")
            gens = _gens()
            losses = []
            for step in range(1, 3):
                clips, me, mp = draw_batch(gens, 4, TINY, 300 + step, 400 + step)
                out = tr.train_step(*to_dev(clips, me, mp), lr=1e-3, wd=0.04, ema=0.99)
                losses.append(out.loss)
            torch.cuda.synchronize()
            if mode != "none":
                assert len(tr.reducer.launched) == len(tr.reducer.buckets) + len(tr.reducer.tail)
            res[mode] = (losses, tr.arena.G.clone().cpu(), tr.arena.P.clone().cpu(), tr.tarena.P.clone().cpu())
        # the C-ABI RCCL binding on its own (not on the trainer's path since round 4): a one-rank communicator created from a
        # unique id, sum-all-reduce and broadcast of a buffer on a side stream leave it unchanged
        import ctypes
        from jepa_amd.hip.lib import check, load_library
        lib = load_library()
        idb = (ctypes.c_ubyte * lib.vj_comm_unique_id_bytes())()
        check(lib.vj_comm_unique_id(idb), "vj_comm_unique_id")
        comm = ctypes.c_void_p()
        check(lib.vj_comm_init(ctypes.byref(comm), 0, 1, bytes(idb)), "vj_comm_init")
        buf = torch.randn(1 << 20, device="cuda")
        ref = buf.clone()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        check(lib.vj_comm_allreduce_bucket(comm, buf.data_ptr(), buf.numel(), st.cuda_stream), "vj_comm_allreduce_bucket")
        check(lib.vj_comm_broadcast(comm, buf.data_ptr(), buf.numel(), 0, st.cuda_stream), "vj_comm_broadcast")
        st.synchronize()
        assert torch.equal(buf, ref)
        check(lib.vj_comm_destroy(comm), "vj_comm_destroy")
        dist.destroy_process_group()
        for mode in ("torch",):
            assert res[mode][0] == res["none"][0], (mode, res[mode][0], res["none"][0])
            for a, b in zip(res[mode][1:], res["none"][1:]):
                assert torch.equal(a, b), mode
        q.put("ok")
    except BaseException as e:   # noqa: BLE001
        import traceback
        q.put(traceback.format_exc() + repr(e))


@pytest.mark.timeout(300)
def test_reducer_at_one_rank_leaves_the_step_bit_identical():
    """with a 1-rank RCCL communicator the bucketed reducer (both backends: torch.distributed and the
    C-ABI vj_comm_*) must leave losses, the gradient arena, the weights and the EMA target BIT-identical to the step without
    a reducer -- a SUM over one rank is the identity, so any difference would be a ordering / stream bug in the bucket path."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_reducer_worker, args=(q,))
    p.start()
    msg = q.get(timeout=280)
    p.join(30)
    assert msg == "ok", msg
