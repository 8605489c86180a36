"""Process-group bring-up and the scalar collectives of the train loop, with the reference's names
(src/utils/distributed.py:18-113): init_distributed, AllReduce (mean), AllReduceSum, AllGather.

One process per GPU over RCCL (torch.distributed backend "nccl" on ROCm); gloo when there is no GPU so the
multi-process paths are testable on CPU.  Rank/world come from torchrun (RANK/WORLD_SIZE), SLURM, or the
explicit pair; like the reference, a failed bring-up degrades to world_size 1.
"""
import os
from logging import getLogger

import torch
import torch.distributed as dist

logger = getLogger()


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def init_distributed(port=37123, rank_and_world_size=(None, None)):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    rank, world_size = rank_and_world_size
    if rank is None or world_size is None:
        if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:          # torchrun / torch.distributed.run
            rank, world_size = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        elif 'SLURM_NTASKS' in os.environ and 'SLURM_PROCID' in os.environ:
            rank, world_size = int(os.environ['SLURM_PROCID']), int(os.environ['SLURM_NTASKS'])
            os.environ.setdefault('MASTER_ADDR', os.environ.get('HOSTNAME', '127.0.0.1'))
        else:
            logger.info('no launcher environment found (distributed training not available)')
            return 1, 0
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(port))
    try:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, world_size=world_size, rank=rank)
    except Exception as e:  # same degradation as the reference (distributed.py:43-45)
        logger.info(f'Rank: {rank}. Distributed training not available {e}')
        return 1, 0
    return world_size, rank


class AllGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if not _active():
            return x
        x = x.contiguous()
        parts = [torch.zeros_like(x) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, x)
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, grads):
        if not _active():
            return grads
        per = grads.shape[0] // dist.get_world_size()
        grads = grads.contiguous()
        dist.all_reduce(grads)
        return grads[per * dist.get_rank():per * (dist.get_rank() + 1)]


class AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if _active():
            x = x.contiguous()
            dist.all_reduce(x)
        return x

    @staticmethod
    def backward(ctx, grads):
        return grads


class AllReduce(torch.autograd.Function):
    """Mean over ranks."""

    @staticmethod
    def forward(ctx, x):
        if _active():
            x = x.contiguous() / dist.get_world_size()
            dist.all_reduce(x)
        return x

    @staticmethod
    def backward(ctx, grads):
        return grads
