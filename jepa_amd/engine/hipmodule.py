"""Glue between nn.Module parameter containers and the HIP chains.

A HipModule owns (or, inside a Trainer, shares) a ParamArena: flat fp32 master weights that the module's
Parameters are views of, with bf16 / transposed-bf16 shadows for the MFMA GEMMs.  Standalone modules build a
private arena lazily on first use and refresh the shadows whenever a parameter's version counter moved (e.g.
after an external optimizer step or load_state_dict).

`run_with_autograd` exposes a whole-module forward as ONE autograd node whose backward is the hand-written
layer chain, so `loss.backward()` + any torch optimizer keep working on these modules.
"""
import torch

from .weights import ARENA_GENERATION, ParamArena, wait_pending_update


def require_gpu(t, what):
    if not t.is_cuda:
        raise ValueError(f"{what}: jepa_amd computes only on the GPU through libvjepa_hip.so; got a CPU tensor "
                         "(there is no CPU fallback)")


class HipModule:
    """Mixin for VisionTransformer / VisionTransformerPredictor."""

    def _hip_attach(self, arena, prefix):
        """Called by the Trainer: parameters live in the trainer's shared arena under `prefix`."""
        self.__dict__["_hip_shared"] = (arena, prefix)

    def _hip_linear_names(self):
        return [n for n, m in self.named_modules() if isinstance(m, torch.nn.Linear)]

    def _hip_arena(self, train):
        wait_pending_update()   # a Trainer's deferred update (overlap_update) must have run before these weights are read
        shared = self.__dict__.get("_hip_shared")
        if shared is not None:
            return shared
        st = self.__dict__.get("_hip_private")
        params = [(n, p) for n, p in self.named_parameters()]
        dev = params[0][1].device
        if dev.type != "cuda":
            raise ValueError("move the module to the GPU first (.to('cuda')): jepa_amd has no CPU compute path")
        first = params[0][1]
        if st is None or st["arena"].device != dev or st["ptr"] != first.data_ptr():
            # weights live in the arena whether or not they require grad (a frozen encoder under eval is the normal
            # inference case, evals/video_classification_frozen/eval.py:414-441); only the sincos position tables --
            # never GEMM operands -- stay outside as fp32 tensors
            tables = ("pos_embed", "predictor_pos_embed")
            weights = [(n, p) for n, p in params if n.split(".")[-1] not in tables]
            arena = ParamArena([weights], dev, with_moments=False, bind_grads=False)
            arena.frozen = {n: p.data.to(torch.float32).contiguous() for n, p in params if n.split(".")[-1] in tables}
            st = {"arena": arena, "ptr": first.data_ptr(), "version": None, "has_T": False,
                  "names": [n for n, _ in weights]}
            self.__dict__["_hip_private"] = st
        arena = st["arena"]
        version = (ARENA_GENERATION[0], sum(p._version for _, p in params))
        if st["version"] != version:
            arena.refresh_bf16()
            if st["has_T"]:
                arena.refresh_transposed()
            st["version"] = version
        if train and not st["has_T"]:
            arena.make_transposed([n + ".weight" for n in self._hip_linear_names()])
            st["has_T"] = True
        return arena, ""

    def _hip_param_list(self):
        arena, prefix = self._hip_arena(train=True)
        named = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
        return arena, prefix, named


class _ModuleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, fwd_fn, bwd_fn, args, n_diff, *tensors):
        arena, prefix, named = module._hip_param_list()
        views = module._hip_views(train=True)
        with torch.no_grad():
            out, extra, saved = fwd_fn(module, views, args, tensors[:n_diff])
        ctx.pack = (module, bwd_fn, views, saved, arena, prefix, [n for n, _ in named], n_diff)
        return out, extra

    @staticmethod
    def backward(ctx, dout, _extra):
        module, bwd_fn, views, saved, arena, prefix, names, n_diff = ctx.pack
        with torch.no_grad():
            in_grads = bwd_fn(module, views, saved, dout.contiguous())
        in_grads = list(in_grads) if in_grads is not None else []
        in_grads += [None] * (n_diff - len(in_grads))
        # Parameters attached to a Trainer already have .grad aliasing the arena slot the chain just wrote: returning a
        # tensor for them would make autograd ADD it onto the same memory (doubling the gradient); return None there.
        pgrads = []
        for n in names:
            g = arena.grad(prefix + n)
            p = arena.slots[prefix + n].param
            aliased = p.grad is not None and p.grad.data_ptr() == g.data_ptr()
            pgrads.append(None if aliased else g.clone())
        return (None, None, None, None, None, *in_grads, *pgrads)


class _Extra:
    """Opaque carrier so non-tensor metadata can ride through autograd.Function outputs."""

    def __init__(self, value):
        self.value = value


def run_with_autograd(module, fwd_fn, bwd_fn, args, diff_inputs=()):
    _, _, named = module._hip_param_list()

    def fwd(m, views, a, diff):
        out, extra, saved = fwd_fn(m, views, a, diff)
        return out, _Extra(extra), saved

    out, extra = _ModuleFn.apply(module, fwd, bwd_fn, args, len(diff_inputs), *diff_inputs, *[p for _, p in named])
    return out, extra.value
