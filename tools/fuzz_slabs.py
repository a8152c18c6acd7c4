#!/usr/bin/env python3
"""Randomised check of the slab layer on ONE GPU: a random cloud is cut into 2-8 x-slabs, every slab runs SlabSearch (halo pack,
[owned | ghosts], candidates-only ghosts, global ids, speculative exchange) in its own thread with the in-process transport of
tests/slab_helpers.py, and the union of the slabs' lists must equal what one engine computes for the whole cloud.

usage: python tools/fuzz_slabs.py [--minutes 5] [--seed 1]       (needs a GPU; test infrastructure, not part of the product)"""
import argparse
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_slabs as TS   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=5.0)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)


def cloud(n, kind):
    if kind == "uniform":
        p = rng.random((n, 3)) * np.array([rng.uniform(1.0, 4.0), 1.0, 1.0])
    elif kind == "clustered":
        k = int(rng.integers(2, 7))
        centres = rng.random((k, 3)) * np.array([3.0, 1.0, 1.0])
        p = centres[rng.integers(0, k, n)] + rng.normal(0.0, 0.08, (n, 3))
    else:   # dam-break like: a dense column at one end, a thin layer over the floor
        m = int(0.7 * n)
        p = np.concatenate([rng.random((m, 3)) * np.array([0.6, 1.0, 1.0]), rng.random((n - m, 3)) * np.array([3.0, 1.0, 0.08])])
    return p.astype(np.float32)


t_end = time.time() + 60.0 * args.minutes
n_scenes = 0
while time.time() < t_end:
    n = int(10.0 ** rng.uniform(3.3, 5.5))
    kind = str(rng.choice(["uniform", "clustered", "dam"]))
    variable = bool(rng.random() < 0.5)
    world = int(rng.integers(2, 9))
    pts = cloud(n, kind)
    r0 = 0.9 * (40.0 / n) ** (1.0 / 3.0) * float(rng.uniform(0.5, 1.0))
    ratio = float(rng.choice([1.5, 3.0]))
    case = types.SimpleNamespace(name="fuzz", points=[pts], radius=None if variable else r0,
                                 radii=[(r0 * (1.0 + (ratio - 1.0) * rng.random(n))).astype(np.float32)] if variable else None,
                                 symmetric=bool(rng.random() < 0.7) if variable else True)
    desc = f"n {n} {kind} world {world} variable {variable} ratio {ratio} r0 {r0:.4g} symmetric {case.symmetric}"
    try:
        s_offs, s_idx = TS._single_device(case)
        if rng.random() < 0.5:     # the C entry points of the slab layer (tnsx_slab_step, in-process transport) ...
            unions, log, sizes, _ = TS._run_slabs_c(case, world, n_steps=int(rng.integers(1, 4)), speculative=bool(rng.random() < 0.8))
            g_offs, g_idx = unions[(0, 0)]
        else:                      # ... or the Python layer over the same protocol
            (g_offs, g_idx), log, sizes, _ = TS._run_slabs(case, world, n_steps=int(rng.integers(1, 4)), speculative=bool(rng.random() < 0.8))
    except ValueError as e:
        if "planes" in str(e) or "slabs asked" in str(e):
            continue            # fewer x planes than slabs: the decomposition refuses, as documented
        raise
    ok = np.array_equal(g_offs, s_offs)
    if ok:
        lid = np.repeat(np.arange(len(g_offs) - 1), np.diff(g_offs))
        a = g_idx[np.lexsort((g_idx, lid))]
        b = np.asarray(s_idx, np.int64)[np.lexsort((np.asarray(s_idx, np.int64), lid))]
        ok = np.array_equal(a, b)
    if not ok:
        print("FAILED: the union of the slabs differs from the single-device result"); print("  scene:", desc); sys.exit(1)
    n_scenes += 1
    if n_scenes % 10 == 0:
        print(f"  {n_scenes} scenes", flush=True)
print(f"slab fuzz ok: {n_scenes} scenes (2-8 slabs, fixed and per-point radii, 1-3 steps, speculative and exact exchange)")
