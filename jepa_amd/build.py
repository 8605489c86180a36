"""Build libvjepa_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

`python -m jepa_amd.build` compiles every csrc/*.hip / *.cpp for --offload-arch=gfx950 (hipcc cross-compiles on a
GPU-less host) and links jepa_amd/lib/libvjepa_hip.so.  Objects are rebuilt only when a source or header is newer.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvjepa_hip.so")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-value", "-ffp-contract=fast"]
# per-file flags: the attention kernels do VALU math (softmax, dS) on MFMA results every tile, so their accumulators
# must live in VGPRs (gfx950 has a unified VGPR/AGPR file): this removes ~1900 v_accvgpr_read/write moves.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the V-JEPA HIP kernels cannot be built on this host")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, obj, headers, extra=()):
    if not _newer(obj, [src, os.path.abspath(__file__)] + headers):
        return obj, False
    cmd = [_hipcc()] + CXXFLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + list(extra) + ["-x", "hip", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, True


def build(verbose=True, force=False, variant=None, extra_flags=()):
    """variant: build an A/B copy `libvjepa_hip_<variant>.so` with extra compiler flags (objects in _build_<variant>);
    selected at run time with VJ_LIB_VARIANT=<variant> (jepa_amd/hip/lib.py).  Experiments only: the product is the
    default library."""
    global OBJ, LIB
    if variant:
        OBJ = os.path.join(CSRC, "_build_" + variant)
        LIB = os.path.join(LIBDIR, f"libvjepa_hip_{variant}.so")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))
    if force:
        shutil.rmtree(OBJ)
        os.makedirs(OBJ)
    jobs = [(s, os.path.join(OBJ, os.path.splitext(os.path.basename(s))[0] + ".o")) for s in srcs]
    rebuilt = False
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        for obj, did in ex.map(lambda j: _compile(j[0], j[1], headers, extra_flags), jobs):
            rebuilt |= did
            if verbose and did:
                print(f"[jepa_amd.build] compiled {os.path.basename(obj)}", flush=True)
    objs = [o for _, o in jobs]
    if rebuilt or _newer(LIB, objs):
        # no rpath: at run time the HIP runtime is the one torch already loaded (same SONAME libamdhip64.so.7)
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[jepa_amd.build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    _args = [a for a in sys.argv[1:] if a != "--force"]
    _variant = _args[0] if _args else None          # python -m jepa_amd.build [variant [extra hipcc flags...]]
    build(force="--force" in sys.argv, variant=_variant, extra_flags=_args[1:])
