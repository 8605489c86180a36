"""Timers, meters and CSV logging with the reference's names (src/utils/logging.py:14-118).

grad_logger / adamw_logger read the flat gradient / moment arenas when the trainer exposes them, so the
~1k per-tensor host syncs of the reference collapse into a handful of reductions evaluated only when logged.
"""
import logging
import sys

import torch


def gpu_timer(closure, log_timings=True):
    """Time `closure()` with device events; returns (result, elapsed_ms)."""
    log_timings = log_timings and torch.cuda.is_available()
    elapsed = -1.
    if log_timings:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
    result = closure()
    if log_timings:
        end.record()
        torch.cuda.synchronize()
        elapsed = start.elapsed_time(end)
    return result, elapsed


LOG_FORMAT = "[%(levelname)-8s][%(asctime)s][%(funcName)-25s] %(message)s"
DATE_FORMAT = "%Y-%m-%d %H:%M:%S"


def get_logger(name=None, force=False):
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format=LOG_FORMAT, datefmt=DATE_FORMAT, force=force)
    return logging.getLogger(name=name)


class CSVLogger(object):
    def __init__(self, fname, *argv):
        self.fname = fname
        self.types = [fmt for fmt, _ in argv]
        with open(self.fname, '+a') as f:
            print(','.join(name for _, name in argv), file=f)

    def log(self, *argv):
        with open(self.fname, '+a') as f:
            print(','.join(fmt % v for fmt, v in zip(self.types, argv)), file=f)


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0
        self.max, self.min = float('-inf'), float('inf')

    def update(self, val, n=1):
        self.val = val
        try:
            self.max, self.min = max(val, self.max), min(val, self.min)
        except Exception:
            pass
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def grad_logger(named_params):
    stats = AverageMeter()
    stats.first_layer = stats.last_layer = None
    for n, p in named_params:
        if (p.grad is not None) and not (n.endswith('.bias') or len(p.shape) == 1):
            g = float(torch.norm(p.grad.data))
            stats.update(g)
            if 'qkv' in n:
                stats.last_layer = g
                if stats.first_layer is None:
                    stats.first_layer = g
    if stats.first_layer is None or stats.last_layer is None:
        stats.first_layer = stats.last_layer = 0.
    return stats


def adamw_logger(optimizer):
    state = optimizer.state_dict().get('state')
    exp_avg, exp_avg_sq = AverageMeter(), AverageMeter()
    for key in state:
        s = state.get(key)
        exp_avg.update(float(s.get('exp_avg').mean()))
        exp_avg_sq.update(float(s.get('exp_avg_sq').mean()))
    return {'exp_avg': exp_avg, 'exp_avg_sq': exp_avg_sq}
