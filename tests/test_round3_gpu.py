"""Round-3 GPU tests: input-edge staging safety, run-time options, the data-parallel reducer at one rank, arena-wide
gradient parity at the benched size, and the kernels added in round 3 (each checked against the kernel it replaces
and against an fp32 PyTorch reference).  Stated tolerances are next to each assertion."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import rel_l2  # noqa: E402
from tests.step_util import TINY, TINY_MASKS, build_trainer, draw_batch, to_dev  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gens(masks=TINY_MASKS, m=TINY):
    from oracle import vjepa_oracle as O
    return O.make_mask_gens(masks, m["crop"], m["frames"], m["patch"], m["tubelet"])


# ------------------------------------------------------------------------------------------------ input edge
def test_prefetcher_pageable_inputs_with_the_host_running_ahead():
    """with pageable loader tensors the prefetcher stages through pinned buffers; a host that runs
    several steps ahead of the GPU (nothing reads the loss) must not overwrite a pinned buffer whose H2D copy has not
    executed yet.  The GPU is kept ~60 ms behind per step by a dummy load; every delivered batch must carry its own
    constant (checked after ONE final synchronise)."""
    from jepa_amd.engine.input import DevicePrefetcher
    n, shape = 8, (8, 3, 16, 112, 112)
    cnt = [0]

    def fetch():
        i = cnt[0]
        cnt[0] += 1
        if i >= n:
            raise StopIteration
        return ([torch.full(shape, float(i))], [torch.full((8, 5), i, dtype=torch.int64)],
                [torch.full((8, 3), i, dtype=torch.int64)])
    pf = DevicePrefetcher(fetch, torch.device(DEV))
    heavy = torch.randn(6144, 6144, device=DEV)
    seen = []
    for i in range(n):
        for _ in range(8):
            heavy @ heavy                       # test-side load only: keeps the device behind the host
        c, me, mp = pf.next()
        seen.append(torch.stack([c.min(), c.max(), me[0].max().float(), mp[0].min().float()]))
    torch.cuda.synchronize()
    for i, s in enumerate(seen):
        assert s.tolist() == [float(i)] * 4, (i, s.tolist())


# ------------------------------------------------------------------------------------------------ run-time options
def test_runtime_options_round_trip_and_unknown_name():
    from jepa_amd.hip.lib import HipKernelError, get_option, set_option
    old = set_option("gemm_4w", 1)
    try:
        assert get_option("gemm_4w") == 1
    finally:
        set_option("gemm_4w", old)
    with pytest.raises(HipKernelError):
        get_option("no_such_option")
