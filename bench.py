#!/usr/bin/env python
"""V-JEPA pretraining-step benchmark on MI355X (BASELINE.json metric: clips/sec, fwd+bwd+EMA, ViT-L/16 16x224x224).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: re-executes itself under
                                                           # torch.distributed.run (one rank per GPU, 127.0.0.1 rendezvous)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # the same ranks started by an external launcher

A "step" is one pass of the hot path (reference app/vjepa/train.py:414-487: schedules, target forward,
context+predictor forward/backward, latent loss, gradient all-reduce, AdamW, EMA) over one batch of synthetic
clips already resident in HBM, with mask indices drawn by the reference-compatible multiblock collator.
Weak scaling: every rank processes its own B=24 clips.  Rank 0 prints ONE JSON line.

`roofline.achieved / frac` is the WHOLE STEP (the north-star quantity): matmul FLOPs of the step (engine/flops.py) over the
timed region, against the dense bf16 MFMA peak.  Its sub-objects `gemm_family`, `dominant_kernel`, `attn_fwd`, `attn_bwd`
come from an instrumented single-stream pass of the same workload run right after the un-instrumented timed region
(HIP events per launch on the launch stream; `value` never includes instrumentation): algorithmic 2*M*N*K of every launch
divided by its duration.  `traffic` = HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes.
`cpu_baseline` times the pinned CPU oracle (a port of the reference arithmetic) on the host cores, rank 0, N=1.
"""
import argparse
import json
import logging
import os
import sys
import time

logging.basicConfig(stream=sys.stderr, level=logging.INFO)   # stdout carries the ONE JSON line and nothing else

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Hardware queues: the PACKAGE decides (jepa_amd/__init__.py sets GPU_MAX_HW_QUEUES=6 unless the user exported a value, and says
# why: with eight queues a second RCCL communicator slows every compute kernel, profiles/r05_dp1_coll_mode.md).  It must be
# imported before torch initialises HIP; the value in force is reported in the JSON line (`config.hw_queues`).
import jepa_amd  # noqa: E402,F401

import torch  # noqa: E402

VITL_MASKS = [  # configs/pretrain/vitl16.yaml:38-62
    dict(aspect_ratio=(0.75, 1.5), num_blocks=8, spatial_scale=(0.15, 0.15), temporal_scale=(1.0, 1.0),
         max_temporal_keep=1.0, max_keep=None),
    dict(aspect_ratio=(0.75, 1.5), num_blocks=2, spatial_scale=(0.7, 0.7), temporal_scale=(1.0, 1.0),
         max_temporal_keep=1.0, max_keep=None),
]
WORKLOADS = {
    "vitl16": dict(model_name="vit_large", crop=224, frames=16, patch=16, tubelet=2, pred_depth=12, pred_dim=384,
                   batch=24, embed_dim=1024, depth=24, masks=VITL_MASKS,
                   desc="V-JEPA pretrain step ViT-L/16 16x224x224, B=24/GPU, 2 multiblock masks (vitl16.yaml)"),
    # BASELINE.json configs[3]: global batch 3072 over 8 GPUs = 384 clips per GPU, walked in micro-batches of 24 inside
    # the step (gradient accumulation; the collator still draws masks for all 384)
    "vith16": dict(model_name="vit_huge", crop=224, frames=16, patch=16, tubelet=2, pred_depth=12, pred_dim=384,
                   batch=384, micro_batch=24, embed_dim=1280, depth=32, masks=VITL_MASKS, distinct_batches=2,
                   desc="V-JEPA pretrain step ViT-H/16 16x224x224, B=384/GPU in micro-batches of 24, 2 multiblock "
                        "masks (vith16.yaml, global batch 3072 at dp8)"),
    # BASELINE.json configs[4]: configs/pretrain/vith16_384.yaml (batch_size 10, crop 384 -> 8x24x24 = 4608 tokens)
    "vith16_384": dict(model_name="vit_huge", crop=384, frames=16, patch=16, tubelet=2, pred_depth=12, pred_dim=384,
                       batch=10, embed_dim=1280, depth=32, masks=VITL_MASKS, distinct_batches=4,
                       desc="V-JEPA pretrain step ViT-H/16 16x384x384 (4608 tokens), B=10/GPU, 2 multiblock masks "
                            "(vith16_384.yaml)"),
    "vittiny": dict(model_name="vit_tiny", crop=64, frames=8, patch=16, tubelet=2, pred_depth=2, pred_dim=96,
                    batch=2, embed_dim=192, depth=12, masks=VITL_MASKS[:1],
                    desc="V-JEPA pretrain step ViT-Tiny/16 8x64x64, B=2, 1 mask (plumbing config)"),
}
HP = dict(ipe=300, ipe_scale=1.25, epochs=300, warmup=40, start_lr=2e-4, lr=6.25e-4, final_lr=1e-6, wd=0.04,
          final_wd=0.4, ema=(0.998, 1.0), betas=(0.9, 0.999), eps=1e-8, loss_exp=1.0, reg_coeff=0.0)
MFMA_BF16_PEAK = 2.5e15  # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md chip table


def build(wl, device, world_size):
    import copy
    from jepa_amd.app.vjepa.utils import init_opt, init_video_model
    torch.manual_seed(0)
    enc, pred = init_video_model(device="cpu", patch_size=wl["patch"], num_frames=wl["frames"],
                                 tubelet_size=wl["tubelet"], model_name=wl["model_name"], crop_size=wl["crop"],
                                 pred_depth=wl["pred_depth"], pred_embed_dim=wl["pred_dim"], uniform_power=True,
                                 use_mask_tokens=True, num_mask_tokens=len(wl["masks"]), zero_init_mask_tokens=True,
                                 use_sdpa=True)
    tgt = copy.deepcopy(enc)
    for p in tgt.parameters():
        p.requires_grad = False
    trainer, _, sched, wd_sched = init_opt(
        encoder=enc, predictor=pred, target_encoder=tgt, wd=HP["wd"], final_wd=HP["final_wd"],
        start_lr=HP["start_lr"], ref_lr=HP["lr"], final_lr=HP["final_lr"], iterations_per_epoch=HP["ipe"],
        warmup=HP["warmup"], num_epochs=HP["epochs"], ipe_scale=HP["ipe_scale"], mixed_precision=True,
        betas=HP["betas"], eps=HP["eps"], loss_exp=HP["loss_exp"], reg_coeff=HP["reg_coeff"], clip_grad=10.0,
        world_size=world_size, device=device, micro_batch=wl.get("micro_batch"),
        overlap_update=os.environ.get("VJ_OVERLAP_UPDATE", "1") != "0")   # as app/vjepa/train.py runs it (optimization.overlap_update)
    return trainer, sched, wd_sched


def make_inputs(wl, n_steps, rank, device, host=False):
    """Synthetic batches resident in HBM (host=True: in pinned host memory, for the --h2d input-edge run).  Large
    workloads keep only `distinct_batches` different batches and cycle through them."""
    from jepa_amd.src.masks.multiblock3d import MaskCollator
    coll = MaskCollator(cfgs_mask=wl["masks"], crop_size=wl["crop"], num_frames=wl["frames"],
                        patch_size=wl["patch"], tubelet_size=wl["tubelet"])
    B = wl["batch"]
    batches = []
    gen = torch.Generator(device=device)
    for step in range(min(n_steps, wl.get("distinct_batches", n_steps))):
        gen.manual_seed(1234 + step + 1000 * rank)
        clips = torch.randn(B, 3, wl["frames"], wl["crop"], wl["crop"], device=device, generator=gen)  # never zeros
        torch.manual_seed(4321 + step + 1000 * rank)
        _, me, mp = coll([(torch.zeros(1), 0) for _ in range(B)])
        if host:
            batches.append((clips.cpu().pin_memory(), [m.pin_memory() for m in me], [m.pin_memory() for m in mp]))
        else:
            batches.append((clips, [m.to(device) for m in me], [m.to(device) for m in mp]))
    return batches


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline_subprocess(wl_name, budget_s=240):
    """Run the CPU leg in its own process with a hard wall-clock bound so the default bench always finishes."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", wl_name],
                           capture_output=True, text=True, timeout=budget_s)
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port",
                "sample": f"cpu leg failed: {r.stderr[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port",
                "sample": f"cpu leg exceeded its {budget_s}s budget on this host"}


def cpu_baseline(wl_name):
    """CPU baseline on a bounded sample of the same workload (B=2, 1 warm-up + 3 timed steps): the REAL reference when
    /root/reference is present (build container; kind "reference", oracle/ref_cpu_step.py), otherwise -- on the GPU box,
    where the reference cannot travel -- the pinned oracle, a CPU port of the same arithmetic (kind "port").  Both were
    timed side by side on the build container: profiles/r03_cpu_reference.json (0.183 vs 0.175 clips/s, 8 threads)."""
    from oracle import ref_cpu_step
    wl = WORKLOADS[wl_name]
    cores = min(os.cpu_count(), 64)   # torch CPU kernels stop scaling (and thrash) far below 256 threads
    if ref_cpu_step.available() and os.environ.get("VJ_CPU_BASELINE_KIND", "") != "port":   # =port: time the oracle beside the reference
        n_timed = 3 if wl_name != "vittiny" else 10
        v, times, threads = ref_cpu_step.time_reference(wl, HP, batch=2, timed=n_timed, threads=cores, log=log)
        return {"value": round(v, 4), "unit": "clips/s", "cores": threads, "kind": "reference",
                "sample": f"facebookresearch/jepa modules + init_opt AdamW + EMA (fp32) on the host cores, same model/masks, "
                          f"B=2, 1 warm-up + {n_timed} timed steps, torch CPU kernels with {threads} threads"}
    from oracle import vjepa_oracle as O
    torch.set_num_threads(cores)
    B = 2
    from jepa_amd.app.vjepa.utils import init_video_model
    torch.manual_seed(0)
    enc, pred = init_video_model(device="cpu", patch_size=wl["patch"], num_frames=wl["frames"],
                                 tubelet_size=wl["tubelet"], model_name=wl["model_name"], crop_size=wl["crop"],
                                 pred_depth=wl["pred_depth"], pred_embed_dim=wl["pred_dim"], uniform_power=True,
                                 use_mask_tokens=True, num_mask_tokens=len(wl["masks"]), zero_init_mask_tokens=True)
    heads = enc.backbone.num_heads
    cfg = dict(embed_dim=wl["embed_dim"], depth=wl["depth"], heads=heads, pred_dim=wl["pred_dim"],
               pred_depth=wl["pred_depth"], num_mask_tokens=len(wl["masks"]), patch=wl["patch"],
               tubelet=wl["tubelet"], num_patches=enc.backbone.num_patches)
    ew = {k[len("backbone."):]: v.detach().clone() for k, v in enc.state_dict().items()}
    pw = {k[len("backbone."):]: v.detach().clone() for k, v in pred.state_dict().items()}
    state = dict(enc=ew, pred=pw, tgt={k: v.clone() for k, v in ew.items()}, opt={})
    gens = O.make_mask_gens(wl["masks"], wl["crop"], wl["frames"], wl["patch"], wl["tubelet"])
    hp = dict(HP)
    times = []
    n_timed = 3 if wl_name != "vittiny" else 10
    log(f"cpu baseline: model built, {cores} threads")
    for step in range(1, 2 + n_timed):
        clips = torch.randn(B, 3, wl["frames"], wl["crop"], wl["crop"], generator=torch.Generator().manual_seed(step))
        torch.manual_seed(4321 + step)
        me, mp = zip(*[g(B) for g in gens])
        t0 = time.time()
        O.train_step(state, clips, list(me), list(mp), cfg, hp, step)
        log(f"cpu baseline: step {step} took {time.time() - t0:.2f}s")
        if step > 1:
            times.append(time.time() - t0)
    v = B / (sum(times) / len(times))
    return {"value": round(v, 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU fp32 restatement of train.py:414-487), same model/masks, B={B}, 1 warm-up + "
                      f"{n_timed} timed steps, torch CPU kernels with {cores} threads"}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves, one process per GPU,
    by re-executing this file under torch.distributed.run (the reference's launcher spawns one process per device the same
    way, app/main.py:28-71).  Rank 0 of the children prints the ONE JSON line on the inherited stdout.  Returns when there
    is nothing to do (N = 1, or the ranks were already started by torchrun / the driver: WORLD_SIZE is set)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if args.stub_step_ms is None:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: device {n_dev} not present ({n_dev} GPU(s) visible on this node)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def timed_loop(run, sync, warmup, steps):
    """W untimed warm-up steps, then EXACTLY K steps between two (barrier + device synchronise) fences.
    Returns (elapsed seconds on this rank, host enqueue seconds, seconds to enqueue the first timed step, last result)."""
    last = None
    for i in range(warmup):
        last = run(i)
    sync()
    log("warm-up done; timing")
    t0 = time.perf_counter()
    t_first = None
    for i in range(warmup, warmup + steps):
        last = run(i)
        if t_first is None:   # the first step is enqueued against an idle GPU: pure host cost, no queue back-pressure
            t_first = time.perf_counter() - t0
    host_enqueue = time.perf_counter() - t0
    sync()
    return time.perf_counter() - t0, host_enqueue, t_first, last


def max_over_ranks(x, world, device):
    if world <= 1:
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def per_rank(x, world, device):
    """[x of rank 0, ..., x of rank world-1] on every rank (the scaling line reports the spread: with per-rank mask draws the slowest
    rank sets the pace, and a max alone cannot tell that skew from communication)."""
    if world <= 1 or not torch.distributed.is_initialized():
        return [float(x)]
    t = torch.tensor([x], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    torch.distributed.all_gather(out, t)
    return [float(o.item()) for o in out]


def spread(vals, scale=1.0, nd=3):
    return {"min": round(min(vals) * scale, nd), "mean": round(sum(vals) / len(vals) * scale, nd), "max": round(max(vals) * scale, nd)}


def stub_main(args):
    """Launcher rehearsal without a GPU (tests/test_bench_launch.py): the same argument handling, self-launch, rendezvous,
    fences, max-over-ranks timing and one-line report as the real run, over a gloo group, with the step replaced by a
    sleep of --stub-step-ms plus one small all-reduce.  `data` says "stub": this line is never a measurement."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    B = args.batch or WORKLOADS[args.workload]["batch"]
    buf = torch.ones(64)

    waited = [0.0]

    def run(i):
        time.sleep(args.stub_step_ms * 1e-3 * (1.0 + 0.5 * rank))   # rank r is slower: the report must carry the MAX
        if world > 1:
            t_c = time.perf_counter()
            torch.distributed.all_reduce(buf)     # a fast rank waits here for the slow one: "exposed communication" of the stub
            if i >= args.warmup:
                waited[0] += time.perf_counter() - t_c
        return i

    def sync():
        if world > 1:
            torch.distributed.barrier()

    elapsed, _, _, _ = timed_loop(run, sync, args.warmup, args.steps)
    ranks_ms = [1e3 * e / args.steps for e in per_rank(elapsed, world, torch.device("cpu"))]
    exs = per_rank(1e3 * waited[0] / args.steps, world, torch.device("cpu"))
    elapsed = max_over_ranks(elapsed, world, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({
            "dp": {"ranks": world, "backend": "gloo (stub)", "route": "blocking", "collectives": "stub",
                   "collective_stream_check": None, "rank_ms_per_step": spread(ranks_ms), "rank_exposed_comm_ms": spread(exs),
                   "rank_compute_ms_per_step": spread([a - b for a, b in zip(ranks_ms, exs)])},
            "metric": "V-JEPA pretrain clips/sec (fwd+bwd+EMA)", "value": round(B * world * args.steps / elapsed, 3),
            "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "stub",
            "config": {"workload": "launcher rehearsal (sleep)", "global_batch": B * world, "per_gpu_batch": B,
                       "parallelism": f"dp{world}"}, "roofline": None, "cpu_baseline": None}), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="vitl16", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override (default: the recipe's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-pass", action="store_true")
    ap.add_argument("--h2d", action="store_true",
                    help="input-edge run: batches start in pinned HOST memory and go through the double-buffered "
                         "prefetcher (engine/input.py); reported on stderr, never as `value`")
    ap.add_argument("--gemm-csv", default=None, help="write one line per GEMM / attention launch of the roofline pass")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--stub-step-ms", type=float, default=None, help=argparse.SUPPRESS)   # launcher test: see stub_main()
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.workload)), flush=True)
        return
    self_launch(args)            # --gpus N > 1 outside a launcher: does not return
    if args.stub_step_ms is not None:
        return stub_main(args)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the V-JEPA step runs only in libvjepa_hip.so (no CPU path)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: device {local_rank} not present ({torch.cuda.device_count()} GPU(s) visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("VJ_FORCE_DP", "0") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from jepa_amd.engine import dp
    from jepa_amd.engine.flops import step_flops
    from jepa_amd.hip import ops

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["batch"] = args.batch
        if wl.get("micro_batch") and wl["micro_batch"] >= args.batch:
            wl["micro_batch"] = None
    log(f"building {wl['model_name']} (CPU init, seed 0) ...")
    trainer, sched, wd_sched = build(wl, device, world)
    log("trainer ready; generating synthetic inputs")
    dp.broadcast_parameters(trainer.arena, trainer.tarena)
    if world > 1:
        trainer.sync_shadows()
    n_total = args.warmup + args.steps
    batches = make_inputs(wl, n_total, rank, device)
    ema0, ema1 = HP["ema"]
    mom = [ema0 + i * (ema1 - ema0) / (HP["ipe"] * HP["epochs"] * HP["ipe_scale"]) for i in range(n_total + 8)]

    def run(i):
        clips, me, mp = batches[i % len(batches)]
        return trainer.train_step(clips, me, mp, lr=sched.step(), wd=wd_sched.step(), ema=mom[i])

    if world > 1 or trainer.reducer.enabled:
        log(f"data parallel: {torch.distributed.get_world_size()} RCCL rank(s), "
            f"{len(trainer.reducer.buckets)} layer buckets + {len(trainer.reducer.tail)} tail range(s)")

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    elapsed, host_enqueue, t_first, last = timed_loop(run, sync, args.warmup, args.steps)
    ranks_ms = [1e3 * e / args.steps for e in per_rank(elapsed, world, device)]   # every rank's own clock around the same K steps
    log(f"host enqueue time {1e3 * host_enqueue / args.steps:.1f} ms/step incl. queue back-pressure, "
        f"{1e3 * t_first:.1f} ms for the first step (GPU step {1e3 * elapsed / args.steps:.1f} ms)")
    dp_info = None
    if trainer.reducer.enabled:   # self-diagnosis of the scaling run: how long the compute stream waited on RCCL
        ex = trainer.reducer.exposed_ms()
        if ex is not None:
            log(f"exposed communication (compute stream blocked in reducer.finish): {ex:.2f} ms/step over the last "
                f"{trainer.reducer.exposed_samples} steps")
            ex = max_over_ranks(ex, world, device)
        from jepa_amd.engine import layers as _layers
        log(f"collectives issued {trainer.reducer.coll_mode!r}; stream picks (candidate index, concurrent with each stream it must not share a "
            f"hardware queue with): {_layers._INDEP_LOG}")
        route = trainer.reducer.route     # the route the buckets actually took (a failed stream check switches sync -> capi)
        exs = per_rank(trainer.reducer.exposed_ms() or 0.0, world, device)
        dp_info = {"ranks": world, "backend": "rccl via the C ABI (vj_comm_*)" if route == "capi" else "rccl via torch.distributed",
                   "collectives": trainer.reducer.coll_mode, "route": route,
                   "collective_stream_check": trainer.reducer.coll_check,
                   "rank_ms_per_step": spread(ranks_ms), "rank_exposed_comm_ms": spread(exs),
                   # a rank's own work = its step time minus what its compute stream waited for the collectives (a fast rank waits there
                   # for the slowest one's mask draw): a wide spread HERE is skew, a large exposed time on EVERY rank is communication
                   "rank_compute_ms_per_step": spread([a - b for a, b in zip(ranks_ms, exs)]),
                   "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "layer_buckets": len(trainer.reducer.buckets), "tail_ranges": len(trainer.reducer.tail),
                   "grad_bytes_per_step": int(trainer.arena.total * 4),
                   "exposed_comm_ms_per_step": None if ex is None else round(ex, 3),
                   "exposed_comm_note": "time the compute stream waited for the gradient collectives in reducer.finish(), "
                                        "max over ranks, mean of the last steps; DESIGN.md section 7 budgets <= 2 ms"}
    elapsed = max_over_ranks(elapsed, world, device)
    loss = last.loss
    B = wl["batch"]
    log(f"timed region: {elapsed:.3f}s for {args.steps} steps -> {B * world * args.steps / elapsed:.2f} clips/s, loss {loss:.5f}")
    kpe = 3 * wl["tubelet"] * wl["patch"] ** 2
    N = (wl["frames"] // wl["tubelet"]) * (wl["crop"] // wl["patch"]) ** 2
    fl = 0.0
    for i in range(args.warmup, n_total):
        _, me, mp = batches[i % len(batches)]
        fl += step_flops(wl["embed_dim"], wl["depth"], wl["pred_dim"], wl["pred_depth"], N, kpe, B,
                         [m.shape[1] for m in me], [m.shape[1] for m in mp])
    step_tflops = fl / elapsed / 1e12  # per GPU

    if os.environ.get("VJ_PHASE_TIMING"):   # diagnostics: where the main stream spends the step (after `value`)
        trainer.phase_events = []
        n_ph = 3
        for i in range(n_ph):
            run(args.warmup + i)
        sync()
        evs, trainer.phase_events = trainer.phase_events, None
        acc = {}
        for (n0, e0), (n1, e1) in zip(evs[:-1], evs[1:]):
            key = n1 if n1 != "start" else "(between steps)"
            acc[key] = acc.get(key, 0.0) + e0.elapsed_time(e1) / n_ph
        log("main-stream phases, ms/step: " + json.dumps({k: round(v, 2) for k, v in acc.items()}))

    input_edge = None
    if args.h2d:   # input edge: batches start in pinned host memory, copied one step ahead
        from jepa_amd.engine.input import DevicePrefetcher
        hb = make_inputs(dict(wl, distinct_batches=min(4, wl.get("distinct_batches", 4))), 4, rank, device, host=True)
        cnt = [0]

        def fetch():
            c, me, mp = hb[cnt[0] % len(hb)]
            cnt[0] += 1
            return [c], me, mp
        pf = DevicePrefetcher(fetch, device)
        n_h = max(3, min(args.steps, 10))

        def run_h(i):
            c, me, mp = pf.next()
            return trainer.train_step(c, me, mp, lr=sched.step(), wd=wd_sched.step(), ema=mom[i % len(mom)])
        for i in range(2):
            run_h(i)
        sync()
        b0 = pf.bytes_copied
        t0 = time.perf_counter()
        for i in range(n_h):
            run_h(i)
        sync()
        dt = time.perf_counter() - t0
        input_edge = {"ms_per_step": round(1e3 * dt / n_h, 3), "clips_per_s": round(B * n_h / dt, 2),
                      "h2d_bytes_per_step": int((pf.bytes_copied - b0) / n_h), "resident_ms_per_step":
                      round(1e3 * elapsed / args.steps, 3),
                      "note": "pinned host batches -> double-buffered copy stream -> step; never reported as value"}
        log(f"input-edge run (PCIe-inclusive): {json.dumps(input_edge)}")

    roof = None
    if not args.no_roofline_pass:
        from jepa_amd.engine import chain
        from jepa_amd.engine.layers import side_stream
        side = side_stream(device)
        side.enabled = False       # per-kernel durations are only meaningful when kernels run one at a time
        ops.KERNEL_TIMERS = {}     # launches issued from Python (patch / predictor embed+proj GEMMs and their wgrads)
        chain.prof_enable(True)    # launches issued by the C chains (every transformer block)
        n_inst = min(3, args.steps)
        for i in range(n_inst):
            run(args.warmup + i)
        sync()
        side.enabled = True
        chain.prof_enable(False)
        timers, ops.KERNEL_TIMERS = ops.KERNEL_TIMERS, None
        import tempfile
        csv_path = args.gemm_csv or os.path.join(tempfile.gettempdir(), f"vj_bench_launches_{os.getpid()}.csv")
        fam = chain.prof_collect(csv_path)
        # per-epilogue split of the C chains' GEMM launches (csv: family,tag,m,n,k,us; family 0 = GEMM, tag 0 = bf16 epilogue
        # with optional bias / residual = gemm_nt_4phase_persist_pre_kernel<0, 4> (round 5: the persistent kernel with the pipelined epilogue),
        # the kernel with the largest share of the step)
        dom = dict(launches=0, ms=0.0, flop=0.0, bytes=0.0)
        with open(csv_path) as fcsv:
            next(fcsv)
            for ln in fcsv:
                f_, tag, m, n, k, us = ln.split(",")[:6]
                if f_ == "0" and tag == "0" and int(n) % 256 != 128:   # (N % 256 == 128 runs gemm_nt_4phase_persist_half_kernel since round 6)
                    m, n, k = int(m), int(n), int(k)
                    dom["launches"] += 1
                    dom["ms"] += float(us) * 1e-3
                    dom["flop"] += 2.0 * m * n * k
                    dom["bytes"] += 2.0 * (m * k + n * k + m * n)   # A and B read once, C written once (bf16)
        if not args.gemm_csv:
            os.unlink(csv_path)
        for name, evs in timers.items():
            f = fam.setdefault(name, dict(launches=0, ms=0.0, flop=0.0))
            f["launches"] += len(evs)
            f["ms"] += sum(s.elapsed_time(e) for s, e, _ in evs)
            f["flop"] += sum(w for _, _, w in evs)
        g = fam["gemm_nt"]
        ach = g["flop"] / (g["ms"] * 1e-3)
        log(f"roofline pass: {json.dumps({k: dict(v, tflops=round(v['flop'] / v['ms'] / 1e9, 1)) for k, v in fam.items()})}")
        # HBM bytes per launch of the dominant kernel: measured with rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
        # correction + WRITE_SIZE), committed under profiles/ -- counters cannot be collected from inside this process
        DOM = "gemm_nt_4phase_persist_pre_kernel<0, 4>"
        traffic, traffic_src = None, None
        for pmc_name in ("r06_hbm_pmc.json", "r05_hbm_pmc.json", "r04_hbm_pmc.json"):   # newest first
            pmc = os.path.join(ROOT, "profiles", pmc_name)
            if os.path.exists(pmc) and args.workload == "vitl16":
                with open(pmc) as fjs:
                    tab = json.load(fjs)
                ent = tab.get(DOM) or tab.get("gemm_nt_4phase_persist_kernel<0, 0>") or tab.get("gemm_nt_4phase_persist_kernel<0>")   # (older tables: the same kernel before its epilogue was pipelined)
                if ent:
                    traffic, traffic_src = round(ent["hbm_bytes_per_launch"]), "profiles/" + pmc_name
                    break
        dom_ach = dom["flop"] / (dom["ms"] * 1e-3) if dom["ms"] > 0 else 0.0
        dom_bytes = dom["bytes"] / dom["launches"] if dom["launches"] else None
        whole = {"achieved": round(step_tflops, 2), "frac": round(step_tflops * 1e12 / MFMA_BF16_PEAK, 4),
                 "tflop_per_clip": round(fl / (args.steps * B) / 1e12, 3)}
        roof = {"bound": "mfma",
                "kernel": "whole V-JEPA step (matmul FLOPs of target forward + context / predictor forward + backward over the "
                          "timed region, two-stream execution); sub-objects from the instrumented single-stream pass",
                "achieved": whole["achieved"], "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": whole["frac"],
                "traffic": traffic, "traffic_unit": f"HBM bytes per launch of {DOM} (PMC)", "traffic_source": traffic_src,
                "whole_step": whole,
                "gemm_family": {"kernels": "gemm_nt_4phase_persist_pre_kernel<0|1|2, 4> (persistent 256x256 NT, two staggered wave groups, two "
                                           "32-MFMA sections per K-tile, cross-tile LDS-DMA prefetch, pipelined epilogue; N % 256 == 128: gemm_nt_4phase_persist_half_kernel), gemm_tn_8phase_kernel "
                                           "(transpose-free weight gradients, the four of a block in one launch); MFMA 16x16x32 bf16",
                                "achieved": round(ach / 1e12, 2), "frac": round(ach / MFMA_BF16_PEAK, 4),
                                "launches_per_step": g["launches"] // n_inst,
                                "avg_launch_us": round(1e3 * g["ms"] / g["launches"], 2),
                                "ms_per_step": round(g["ms"] / n_inst, 2)},
                "dominant_kernel": {"name": DOM, "achieved": round(dom_ach / 1e12, 2),
                                    "frac": round(dom_ach / MFMA_BF16_PEAK, 4),
                                    "launches_per_step": dom["launches"] // n_inst,
                                    "avg_launch_us": round(1e3 * dom["ms"] / max(dom["launches"], 1), 2),
                                    "ms_per_step": round(dom["ms"] / n_inst, 2),
                                    "algorithmic_bytes_per_launch": None if dom_bytes is None else round(dom_bytes),
                                    "traffic_bytes_per_launch": traffic,
                                    "traffic_over_algorithmic": (round(traffic / dom_bytes, 3)
                                                                 if traffic and dom_bytes else None)},
                "attn_fwd": {"tflops": round(fam["attn_fwd"]["flop"] / fam["attn_fwd"]["ms"] / 1e9, 1),
                             "ms_per_step": round(fam["attn_fwd"]["ms"] / n_inst, 2)},
                "attn_bwd": {"tflops": round(fam["attn_bwd"]["flop"] / fam["attn_bwd"]["ms"] / 1e9, 1),
                             "ms_per_step": round(fam["attn_bwd"]["ms"] / n_inst, 2)}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline leg (bounded subprocess)")
        cpu = cpu_baseline_subprocess(args.workload)

    if rank == 0:
        line = {
            "metric": "V-JEPA pretrain clips/sec (fwd+bwd+EMA)", "value": round(B * world * args.steps / elapsed, 3),
            "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl["desc"], "global_batch": B * world, "per_gpu_batch": B,
                       "parallelism": f"dp{world}", "final_loss": round(loss, 6),
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if dp_info is not None:
            line["dp"] = dp_info
        if input_edge is not None:
            line["input_edge"] = input_edge
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
