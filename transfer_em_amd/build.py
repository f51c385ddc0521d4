"""Build recipe of libtem_hip.so: hipcc, gfx950 only, in-tree output (transfer_em_amd/lib/).

Every csrc/*.hip is compiled to its own object (in parallel, only when it or a header changed), then linked."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libtem_hip.so")
OBJ = os.path.join(HERE, "lib", "obj")
SOURCES = ["conv_direct.hip", "conv_bww.hip", "bww_lds.hip", "conv_lds.hip", "elementwise.hip", "dispatch.hip",
           "stencil_c1.hip", "convT_mfma.hip", "datapipe.hip", "conv_bf16.hip", "conv3_bf16.hip", "convT_bf16.hip", "bww_bf16.hip", "elementwise_bf16.hip", "wino.hip", "wino_bww.hip", "bww_c1.hip", "conv_s2.hip", "c1out_mfma.hip", "bww_s2.hip", "head.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-pass-failed"] + os.environ.get("TEM_BUILD_FLAGS", "").split()
# TEM_BUILD_FLAGS=-DTEM_DEBUG_KNOBS: the environment knobs of csrc/tem_common.h (microbenchmarks only; off in the shipped build)


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(HERE, "..", "include", "tem_hip.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


FLAGS_STAMP = os.path.join(HERE, "lib", ".build_flags")     # the flags the objects were built with: a knob build is never mistaken for the shipped one


def _flags_changed():
    try:
        return open(FLAGS_STAMP).read() != " ".join(FLAGS)
    except OSError:
        return True


def needs_build():
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    return _flags_changed() or _stale(OUT, srcs + _headers())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    force = force or _flags_changed()
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if not os.path.exists(src):
            continue
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    with open(FLAGS_STAMP, "w") as f:
        f.write(" ".join(FLAGS))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
