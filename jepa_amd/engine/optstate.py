"""torch.optim.AdamW-compatible (de)serialisation of the flat Adam-moment arenas.

The reference checkpoints `optimizer.state_dict()` under the key 'opt' (app/vjepa/train.py:331-342) where the optimizer
is `torch.optim.AdamW(param_groups)` with the four groups of init_opt (app/vjepa/utils.py:173-191), built from ALL
named parameters -- the frozen sincos tables (`pos_embed`, `predictor_pos_embed`) are members of groups 0 and 1 and
consume a parameter id although they never get optimizer state.  These helpers are pure (any device, no kernels), so
the id / shape contract is unit-tested on CPU against a real torch.optim.AdamW built the reference's way.
"""
from dataclasses import dataclass

import torch

ALIGN = 64


def pad64(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


def is_no_decay(name, p):
    """Group split of init_opt (reference app/vjepa/utils.py:173-191)."""
    return ("bias" in name) or (p.dim() == 1)


@dataclass
class Slot:
    name: str
    off: int
    numel: int
    shape: tuple
    param: torch.nn.Parameter


def layout(groups):
    """Arena layout shared by the master / gradient / moment arenas: `groups` = list of lists of (name, param); every
    tensor is padded to 64 elements, every group is one contiguous range.  -> (slots {name: Slot}, group_ranges, total)"""
    slots, ranges, off = {}, [], 0
    for grp in groups:
        start = off
        for name, p in grp:
            slots[name] = Slot(name, off, p.numel(), tuple(p.shape), p)
            off += pad64(p.numel())
        ranges.append((start, off))
    return slots, ranges, off


def reference_groups(enc_named, pred_named):
    """[(group dict, [(name, param), ...])] in the reference's order: enc decayed, pred decayed, enc no-decay (bias /
    1-D, weight_decay 0), pred no-decay.  *_named: iterables of (name, parameter) = module.named_parameters()."""
    enc_named, pred_named = list(enc_named), list(pred_named)

    def grp(named, nodecay):
        return [(n, p) for n, p in named if is_no_decay(n, p) == nodecay]
    return [({}, grp(enc_named, False)), ({}, grp(pred_named, False)),
            ({"WD_exclude": True, "weight_decay": 0}, grp(enc_named, True)),
            ({"WD_exclude": True, "weight_decay": 0}, grp(pred_named, True))]


def build_state_dict(param_groups, slot_of, M1, M2, step, betas, eps):
    """param_groups: list of dicts with 'params' (parameters, frozen ones included); slot_of: id(param) -> Slot for the
    trainable ones; M1 / M2: flat exp_avg / exp_avg_sq arenas; step: Adam step count t (0 = no state yet)."""
    state, groups, k = {}, [], 0
    for g in param_groups:
        ids = []
        for p in g["params"]:
            s = slot_of.get(id(p))
            if s is not None and step > 0:   # torch creates state lazily at the first step; frozen params never get one
                state[k] = {"step": torch.tensor(float(step)),
                            "exp_avg": M1[s.off:s.off + s.numel].view(s.shape).clone(),
                            "exp_avg_sq": M2[s.off:s.off + s.numel].view(s.shape).clone()}
            ids.append(k)
            k += 1
        gd = {kk: v for kk, v in g.items() if kk not in ("params", "_range")}
        gd.update({"params": ids, "betas": tuple(betas), "eps": eps, "amsgrad": False, "maximize": False,
                   "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                   "decoupled_weight_decay": True})
        groups.append(gd)
    return {"state": state, "param_groups": groups}


def load_state_dict(param_groups, slot_of, M1, M2, sd):
    """Inverse of build_state_dict; also accepts the 'opt' entry of a reference checkpoint.  Validates group sizes and
    tensor shapes BEFORE writing anything (a mismatching checkpoint leaves the arenas untouched).  Returns the step
    count found (None when the checkpoint holds no state)."""
    if len(sd["param_groups"]) != len(param_groups):
        raise ValueError(f"optimizer state has {len(sd['param_groups'])} parameter groups, expected {len(param_groups)}")
    todo, step = [], None
    for gi, (g, gs) in enumerate(zip(param_groups, sd["param_groups"])):
        if len(gs["params"]) != len(g["params"]):
            raise ValueError(f"optimizer state group {gi} holds {len(gs['params'])} parameters, expected "
                             f"{len(g['params'])}")
        for p, k in zip(g["params"], gs["params"]):
            st = sd["state"].get(k)
            s = slot_of.get(id(p))
            if st is None or s is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(s.shape):
                raise ValueError(f"optimizer state shape mismatch for {s.name}: {tuple(st['exp_avg'].shape)} vs {s.shape}")
            todo.append((s, st))
    for s, st in todo:
        M1[s.off:s.off + s.numel].copy_(st["exp_avg"].reshape(-1))
        M2[s.off:s.off + s.numel].copy_(st["exp_avg_sq"].reshape(-1))
        step = int(float(st["step"]))
    for g, gs in zip(param_groups, sd["param_groups"]):
        g["lr"] = gs.get("lr", g.get("lr", 0.0))
        g["weight_decay"] = gs.get("weight_decay", g.get("weight_decay", 0.0))
    return step
