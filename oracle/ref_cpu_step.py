"""CPU timing of the REAL reference on the V-JEPA pretraining step (test / measurement infrastructure only).

The reference (facebookresearch/jepa, read-only at /root/reference, present in the build container only) is imported
unmodified: `init_video_model`, `init_opt` (torch.optim.AdamW + schedulers), `MaskCollator`, `apply_masks` and the
wrapped ViT / predictor modules.  Only the local closure app/vjepa/train.py:414-487 -- not importable, it is a nested
function of main() -- is driven from here, exactly as oracle/make_golden.py does for the fixtures.  fp32 on the host
cores (the reference's CPU path: `torch.cuda.amp.autocast` is a no-op without a GPU).

    python oracle/ref_cpu_step.py --workload vitl16 --batch 2 --timed 3   ->  one JSON line on stdout
`bench.py`'s cpu_baseline leg calls `time_reference()` when /root/reference exists (kind: "reference"), and falls back to
the pinned port (oracle/vjepa_oracle.py, kind: "port") on the GPU box, where the reference is absent.
"""
import argparse
import copy
import json
import os
import sys
import time

import torch

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "src", "models"))


def time_reference(wl, hp, batch=2, timed=3, threads=None, log=None):
    """wl / hp: bench.py's WORKLOADS entry and HP dict.  Returns (clips_per_s, per_step_seconds, threads)."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import logging
    logging.disable(logging.INFO)   # the reference logs both module trees at INFO on construction
    # the reference's package names (`src`, `app`) are resolved from /root/reference: our mirror lives under jepa_amd.*
    import torch.nn.functional as F
    from app.vjepa.utils import init_opt, init_video_model
    from src.masks.multiblock3d import MaskCollator
    from src.masks.utils import apply_masks
    threads = threads or min(os.cpu_count(), 64)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    encoder, predictor = init_video_model(
        device="cpu", patch_size=wl["patch"], num_frames=wl["frames"], tubelet_size=wl["tubelet"],
        model_name=wl["model_name"], crop_size=wl["crop"], pred_depth=wl["pred_depth"], pred_embed_dim=wl["pred_dim"],
        uniform_power=True, use_mask_tokens=True, num_mask_tokens=len(wl["masks"]), zero_init_mask_tokens=True,
        use_sdpa=True)
    target = copy.deepcopy(encoder)
    for p in target.parameters():
        p.requires_grad = False
    optimizer, _, scheduler, wd_scheduler = init_opt(
        encoder=encoder, predictor=predictor, wd=hp["wd"], final_wd=hp["final_wd"], start_lr=hp["start_lr"],
        ref_lr=hp["lr"], final_lr=hp["final_lr"], iterations_per_epoch=hp["ipe"], warmup=hp["warmup"],
        num_epochs=hp["epochs"], ipe_scale=hp["ipe_scale"], mixed_precision=False, betas=hp["betas"], eps=hp["eps"])
    collator = MaskCollator(cfgs_mask=wl["masks"], crop_size=wl["crop"], num_frames=wl["frames"],
                            patch_size=wl["patch"], tubelet_size=wl["tubelet"])
    ema0, ema1 = hp["ema"]
    total = hp["ipe"] * hp["epochs"] * hp["ipe_scale"]
    times = []
    for step in range(1, 2 + timed):
        clips = torch.randn(batch, 3, wl["frames"], wl["crop"], wl["crop"], generator=torch.Generator().manual_seed(step))
        torch.manual_seed(4321 + step)
        _, masks_enc, masks_pred = collator([(torch.zeros(1), 0) for _ in range(batch)])
        t0 = time.time()
        # ---- app/vjepa/train.py:414-487, through the reference's own modules ----
        scheduler.step()
        wd_scheduler.step()
        with torch.no_grad():
            h = target(clips)
            h = F.layer_norm(h, (h.size(-1),))
            h = apply_masks(h, masks_pred, concat=False)
        z = predictor(encoder(clips, masks_enc), h, masks_enc, masks_pred)
        loss = 0.
        for zi, hi in zip(z, h):
            loss += torch.mean(torch.abs(zi - hi) ** hp["loss_exp"]) / hp["loss_exp"]
        loss /= len(masks_pred)
        pstd = sum([torch.sqrt(zi.var(dim=1) + 0.0001) for zi in z]) / len(z)
        loss = loss + hp["reg_coeff"] * torch.mean(F.relu(1. - pstd))
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
        m = ema0 + (step - 1) * (ema1 - ema0) / total
        with torch.no_grad():
            for pq, pk in zip(encoder.parameters(), target.parameters()):
                pk.data.mul_(m).add_((1. - m) * pq.detach().data)
        dt = time.time() - t0
        if log:
            log(f"reference CPU step {step}: {dt:.2f}s, loss {float(loss):.5f}")
        if step > 1:
            times.append(dt)
    logging.disable(logging.NOTSET)
    return batch / (sum(times) / len(times)), times, threads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vitl16")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--timed", type=int, default=3)
    ap.add_argument("--threads", type=int, default=None)
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    if not available():
        raise SystemExit("/root/reference is not present on this host")
    v, times, threads = time_reference(bench.WORKLOADS[args.workload], bench.HP, args.batch, args.timed, args.threads,
                                       log=lambda s: print(s, file=sys.stderr, flush=True))
    print(json.dumps({"value": round(v, 4), "unit": "clips/s", "cores": threads, "kind": "reference",
                      "step_seconds": [round(t, 2) for t in times],
                      "sample": f"facebookresearch/jepa modules + init_opt AdamW + EMA on the host cores (fp32), "
                                f"{bench.WORKLOADS[args.workload]['desc']} at B={args.batch}, 1 warm-up + {args.timed} timed steps, "
                                f"{threads} torch threads"}))


if __name__ == "__main__":
    main()
