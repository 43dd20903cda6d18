"""Build libmobody_hip.so (gfx950) in-tree with hipcc.  `python build.py [--force]`."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmobody_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]
FLAGS += os.environ.get("MOBODY_EXTRA_FLAGS", "").split()   # A/B aid: e.g. MOBODY_EXTRA_FLAGS="-DBF_BUFFER_LOADS=0" python build.py --force
if os.environ.get("MOBODY_TRACE") == "1":          # diagnostic build: phase timestamps in the MLP kernels (needs -fgpu-rdc
    FLAGS += ["-DMOBODY_TRACE", "-fgpu-rdc"]       # for the one trace buffer shared by the translation units)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(HERE, "*.h"))) + [os.path.join(HERE, "..", "..", "include", "mobody_hip.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *(["-fgpu-rdc"] if "-fgpu-rdc" in FLAGS else []), "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
