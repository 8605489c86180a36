"""Learning-rate / weight-decay schedules with the reference's classes and return values
(src/utils/schedulers.py:11-76)."""
import math


class WarmupCosineSchedule(object):
    def __init__(self, optimizer, warmup_steps, start_lr, ref_lr, T_max, last_epoch=-1, final_lr=0.):
        self.optimizer = optimizer
        self.start_lr, self.ref_lr, self.final_lr = start_lr, ref_lr, final_lr
        self.warmup_steps = warmup_steps
        self.T_max = T_max - warmup_steps
        self._step = 0.

    def step(self):
        self._step += 1
        if self._step < self.warmup_steps:
            frac = float(self._step) / float(max(1, self.warmup_steps))
            new_lr = self.start_lr + frac * (self.ref_lr - self.start_lr)
        else:
            frac = float(self._step - self.warmup_steps) / float(max(1, self.T_max))
            cosine = 0.5 * (1. + math.cos(math.pi * frac))
            new_lr = max(self.final_lr, self.final_lr + (self.ref_lr - self.final_lr) * cosine)
        for group in self.optimizer.param_groups:
            group['lr'] = new_lr
        return new_lr


class CosineWDSchedule(object):
    def __init__(self, optimizer, ref_wd, T_max, final_wd=0.):
        self.optimizer = optimizer
        self.ref_wd, self.final_wd, self.T_max = ref_wd, final_wd, T_max
        self._step = 0.

    def step(self):
        self._step += 1
        frac = self._step / self.T_max
        new_wd = self.final_wd + (self.ref_wd - self.final_wd) * 0.5 * (1. + math.cos(math.pi * frac))
        new_wd = max(self.final_wd, new_wd) if self.final_wd <= self.ref_wd else min(self.final_wd, new_wd)
        for group in self.optimizer.param_groups:
            if not group.get('WD_exclude', False):
                group['weight_decay'] = new_wd
        return new_wd
