"""Timers, meters and CSV logging with the reference's names (src/utils/logging.py:14-118).

`grad_logger` / `adamw_logger` accept the fused Trainer (engine/step.py): every per-tensor statistic then comes from
ONE `vj_grad_stats_multi` launch over the flat gradient / moment arenas and one small device-to-host copy, evaluated
only when a log line is due -- the reference pays a `float()` host sync per tensor (~1000 per step).  Given plain
named parameters / a torch optimizer (CPU tensors, the oracle) they fall back to per-tensor reductions.
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


def _meter_from(values):
    m = AverageMeter()
    for v in values:
        m.update(v)
    return m


def grad_logger(source, which=None):
    """Norms of the weight-matrix gradients: mean/min/max over tensors plus the first and last qkv layer.
    source: the fused Trainer (with which='enc' | 'pred') or an iterable of (name, parameter)."""
    if hasattr(source, 'arena_stats'):
        rows = [(n, g) for n, g, is_mat in source.arena_stats()[which]['grads'] if is_mat]
    else:
        rows = [(n, float(torch.norm(p.grad.data))) for n, p in source
                if (p.grad is not None) and not (n.endswith('.bias') or len(p.shape) == 1)]
    stats = _meter_from(g for _, g in rows)
    qkv = [g for n, g in rows if 'qkv' in n]
    stats.first_layer, stats.last_layer = (qkv[0], qkv[-1]) if qkv else (0., 0.)
    return stats


def adamw_logger(optimizer):
    """Mean magnitude of the first / second Adam moments per tensor (mean/min/max over tensors)."""
    if hasattr(optimizer, 'arena_stats'):
        st = optimizer.arena_stats()
        rows = st['enc']['moments'] + st['pred']['moments']
        return {'exp_avg': _meter_from(a for a, _ in rows), 'exp_avg_sq': _meter_from(b for _, b in rows)}
    state = optimizer.state_dict().get('state')
    return {'exp_avg': _meter_from(float(s.get('exp_avg').abs().mean()) for s in state.values()),
            'exp_avg_sq': _meter_from(float(s.get('exp_avg_sq').abs().mean()) for s in state.values())}
