"""CPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly what
include/vjepa_hip.h declares (no compute is launched here -- there is no GPU on this host)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vjepa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vj_[a-z0-9_]+)\s*\(", text)) - {"vj_stream_t"})


def test_library_builds_and_exports_every_declared_symbol():
    from jepa_amd import build
    from jepa_amd.hip import lib as L
    build.build(verbose=False)
    lib = L.load_library()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vjepa_hip.h but not exported"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature"
    for n in L.SIGNATURES:
        assert n in names, f"{n} bound in python but not declared in the header"
    assert lib.vj_abi_version() == 1


def test_no_cpu_fallback_in_ops():
    """The product path must fail loudly on CPU tensors instead of silently computing somewhere else."""
    import pytest
    import torch
    from jepa_amd.hip import ops
    x = torch.zeros(4, 8, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.layernorm_fwd(x, torch.ones(8), torch.zeros(8), 1e-6)
    with pytest.raises(ValueError):
        ops.gemm_nt(x, x)
