#!/usr/bin/env python
"""cProfile of the host side of the ViT-L step (enqueue only): python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    wl = dict(bench.WORKLOADS["vitl16"])
    trainer, sched, wd_sched = bench.build(wl, dev, 1)
    batches = bench.make_inputs(wl, 4, 0, dev)

    def run(i):
        clips, me, mp = batches[i % 4]
        return trainer.train_step(clips, me, mp, lr=1e-4, wd=0.04, ema=0.998)
    for i in range(2):
        run(i)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(3):
        run(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
