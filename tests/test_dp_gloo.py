"""Data-parallel plumbing on CPU (gloo, world_size 2): the bucketed gradient reducer walks the same hooks the HIP
backward fires, every element of the gradient arena is summed over ranks exactly once, the bucket order follows
backward completion (predictor first, encoder layers last->first, tail last), parameters broadcast from rank 0,
and init_distributed picks up the torchrun environment."""
import os
import socket
import sys
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_arena(n_enc=3, n_pred=5, D=8, Dp=4):
    """Slots laid out like ParamArena: [enc decayed | enc no-decay | pred decayed | pred no-decay], 64-padded."""
    names = []
    names.append(("enc.patch_embed.proj.weight", D * 6))
    for i in range(n_enc):
        for l, n in (("attn.qkv", 3 * D * D), ("attn.proj", D * D), ("mlp.fc1", 4 * D * D), ("mlp.fc2", 4 * D * D)):
            names.append((f"enc.blocks.{i}.{l}.weight", n))
    names.append(("enc.patch_embed.proj.bias", D))
    for i in range(n_enc):
        for l, n in (("norm1.weight", D), ("norm1.bias", D), ("attn.qkv.bias", 3 * D), ("mlp.fc2.bias", D)):
            names.append((f"enc.blocks.{i}.{l}", n))
    names.append(("pred.predictor_embed.weight", D * Dp))
    names.append(("pred.mask_tokens.0", Dp))
    for i in range(n_pred):
        for l, n in (("attn.qkv", 3 * Dp * Dp), ("attn.proj", Dp * Dp), ("mlp.fc1", 4 * Dp * Dp), ("mlp.fc2", 4 * Dp * Dp)):
            names.append((f"pred.predictor_blocks.{i}.{l}.weight", n))
    names.append(("pred.predictor_proj.weight", D * Dp))
    names.append(("pred.predictor_proj.bias", D))
    slots, off = {}, 0
    for n, k in names:
        slots[n] = SimpleNamespace(off=off, numel=k)
        off += (k + 63) // 64 * 64
    arena = SimpleNamespace(slots=slots, total=off, G=torch.zeros(off), P=torch.zeros(off))
    vit = SimpleNamespace(blocks=[None] * n_enc)
    pred = SimpleNamespace(predictor_blocks=[None] * n_pred)
    return arena, vit, pred


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from jepa_amd.engine import dp
    from jepa_amd.src.utils.distributed import AllReduce, init_distributed
    ws, rk = init_distributed()
    assert (ws, rk) == (world, rank) and dist.get_backend() == "gloo"
    arena, vit, pred = _fake_arena()
    red = dp.GradReducer(arena, vit, pred, world, overlap=True, pred_layers_per_bucket=2)
    # every element is covered by exactly one bucket/tail range
    cover = torch.zeros(arena.total)
    for rs in list(red.buckets.values()) + [red.tail]:
        for lo, hi in rs:
            cover[lo:hi] += 1
    assert bool((cover == 1).all()), "bucket ranges must partition the gradient arena"
    for step in range(2):
        g = torch.Generator().manual_seed(100 * step + rank)
        arena.G.copy_(torch.randn(arena.total, generator=g))
        mine = arena.G.clone()
        red.begin()
        n_pred, n_enc = len(pred.predictor_blocks), len(vit.blocks)
        red.layer_done("pred", n_pred)
        for li in range(n_pred - 1, -1, -1):
            red.layer_done("pred", li)
        red.layer_done("pred", -1)
        for li in range(n_enc - 1, -1, -1):
            red.layer_done("enc", li)
        red.layer_done("enc", -1)
        red.finish()
        others = [torch.randn(arena.total, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)]
        assert torch.allclose(arena.G, sum(others), atol=1e-6)
        assert torch.allclose(others[rank], mine)
        # launch order follows backward completion: predictor proj first, encoder patch embed + tail last
        first = red.launched[0]
        assert first == red.buckets[("pred", n_pred)][0]
        assert red.launched[-len(red.tail):] == red.tail
        enc_last = red.launched.index(red.buckets[("enc", 0)][0])
        enc_first = red.launched.index(red.buckets[("enc", n_enc - 1)][0])
        assert enc_first < enc_last
    # parameter broadcast + scalar mean
    arena.P.fill_(float(rank + 1))
    dp.broadcast_parameters(arena)
    assert bool((arena.P == 1.0).all())
    m = AllReduce.apply(torch.tensor([float(rank)]))
    assert abs(float(m) - (world - 1) / 2) < 1e-6
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


@pytest.mark.timeout(120)
def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_reducer_is_noop_for_world_size_one():
    from jepa_amd.engine import dp
    arena, vit, pred = _fake_arena()
    red = dp.GradReducer(arena, vit, pred, 1)
    red.begin()
    red.layer_done("enc", 0)
    red.finish()
    assert red.launched == [] and not red.enabled
