"""Probe: does this RCCL accept two ranks on ONE GPU (it would let the N > 1 reducer path run on a one-GPU box)?
Answer (round 6, RCCL 2.26.6): no -- "Duplicate GPU detected", also with RCCL_ENABLE_MULTI_RANK_GPU=1 / NCCL_MULTI_RANK_GPU_ENABLE=1.  The two-rank
parity test therefore stays on gloo (tests/test_dp_gpu.py) and the RCCL route is verified at one rank (VJ_FORCE_DP=1).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 lab/probes/rccl_two_ranks_one_gpu.py"""
import os
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
t = torch.full((1024,), float(rank + 1), device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
print(f"rank {rank}: all_reduce -> {t[0].item()} (expected {world * (world + 1) / 2})", flush=True)
dist.destroy_process_group()
