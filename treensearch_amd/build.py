"""Builds treensearch_amd/lib/libtnsx.so (the C-ABI library of include/tnsx.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is git-ignored
but travels to the GPU box with the tree.  `python -m treensearch_amd.build` rebuilds it.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libtnsx.so")
SOURCES = ["tnsx_kernels.hip", "tnsx_build.hip", "tnsx_query.hip", "tnsx_engine.cpp", "tnsx_multi.cpp", "tnsx_slab.cpp"]
HEADERS = ["tnsx_kernels.h", "tnsx_device.h", "tnsx_pool.h", "tnsx_multi.h", os.path.join(ROOT, "include", "tnsx.h")]

# -ffp-contract=off: the neighbour predicate must not be re-associated or fused behind our back
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-x", "hip"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _sources():
    return SOURCES


def _commands():
    """[(source, object, command line)]: everything that decides what an object file contains."""
    extra = os.environ.get("TNSX_EXTRA_FLAGS", "").split()     # experiments, e.g. -DTNSX_FAST_WAVES_PER_EU=5
    out = []
    for src in _sources():
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + ".o")
        out.append((src, obj, [hipcc()] + FLAGS + extra + ["-I" + os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, src), "-o", obj]))
    return out


def _recorded(obj: str) -> str:
    try:
        return open(obj + ".cmd").read()
    except OSError:
        return ""


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in _sources()] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    deps.append(os.path.abspath(__file__))
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    # the library on disk was linked from objects compiled with other flags (an experiment's TNSX_EXTRA_FLAGS)
    return _recorded(LIB) != "\n".join(" ".join(cmd) for _, _, cmd in _commands())


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    objs, procs = [], []
    dep_t = max(os.path.getmtime(h if os.path.isabs(h) else os.path.join(CSRC, h)) for h in HEADERS + [os.path.abspath(__file__)])
    cmds = _commands()
    for src, obj, cmd in cmds:   # one hipcc per translation unit, all at once (the query kernels alone take two minutes)
        objs.append(obj)
        # an object is reused only if it is newer than everything it depends on AND was compiled with exactly this command line (recorded beside it):
        # objects of an experiment (TNSX_EXTRA_FLAGS) never end up in a later plain build
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(dep_t, os.path.getmtime(os.path.join(CSRC, src))) and _recorded(obj) == " ".join(cmd):
            continue
        if os.path.exists(obj + ".cmd"):
            os.remove(obj + ".cmd")
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((obj, cmd, subprocess.Popen(cmd)))
    failed = None
    for obj, cmd, p in procs:
        if failed is not None:      # one compilation failed: do not leave the others running behind the exception
            p.kill()
            p.wait()
            continue
        if p.wait() != 0:
            failed = (p.returncode, cmd)
            continue
        with open(obj + ".cmd", "w") as f:
            f.write(" ".join(cmd))
    if failed is not None:
        raise subprocess.CalledProcessError(failed[0], failed[1])
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(LIB + ".cmd", "w") as f:
        f.write("\n".join(" ".join(c) for _, _, c in cmds))
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
