"""Build the HIP library in-tree:  python -m nsfnet_amd.build [--force]

hipcc cross-compiles gfx950 without a GPU.  Output: nsfnet_amd/lib/libnsfnet_pinn.so
(git-ignored; it travels to the GPU box with the working tree).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libnsfnet_pinn.so")
SOURCES = ["fwd.hip", "bwd.hip", "dw.hip", "fwd_bf16.hip", "bwd_bf16.hip", "dw_bf16.hip", "fwd_wide.hip", "bwd_wide.hip", "dw_wide.hip",
           "fwd_bf16_wide.hip", "bwd_bf16_wide.hip", "dw_bf16_wide.hip", "fwd_bf16_pipe.hip", "bwd_bf16_pipe.hip", "fwd_bf16_split.hip", "bwd_bf16_split.hip", "fwd_bf16_wsplit.hip", "bwd_bf16_wsplit.hip", "misc.hip", "capi.hip"]
HEADERS = ["kernels.h", "layout.h", "bf16_util.h", "reduce_util.h", "point_stage.h", os.path.join("..", "..", "include", "nsfnet_pinn.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file extras.  The pipelined kernels place their epilogue VALU in MFMA shadows: packed-f32 VALU (v_pk_*_f32, what
# the SLP vectoriser makes of adjacent scalar f32 ops) costs more beside MFMAs than the two scalar ops it replaces
# (MI355X_MICROARCH.md, per-instruction constants), and comes with v_mov shuffles.
# -pragma-unroll-threshold: a slot body (64 steps x (6 MFMAs + an epilogue slice)) must unroll completely - register
# arrays are indexed by the step - and is larger than the default cap on `#pragma unroll`.
_PIPE = ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"]
EXTRA_FLAGS = {"fwd_bf16_pipe.hip": _PIPE, "bwd_bf16_pipe.hip": _PIPE, "fwd_bf16_split.hip": _PIPE, "bwd_bf16_split.hip": _PIPE,
               "fwd_bf16_wsplit.hip": _PIPE, "bwd_bf16_wsplit.hip": _PIPE}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [_hipcc()] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr))
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
