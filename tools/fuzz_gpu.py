#!/usr/bin/env python3
"""Randomised comparison of the three ways this repository has of computing the same neighbour lists, for as long as asked:

  A  the engine as shipped (single-pass pool mode: fast / fat / general tier, temporal reuse, optional sorted lists, ...)
  B  the same library with exact_layout=True and temporal_reuse=False (count -> scan -> fill through the general kernel only)
  C  the CPU restatement under oracle/ (all pairs), whenever the scene is small enough

Scenes: 1-3 point sets of random sizes (0 .. ~30 k, now and then up to 200 k, log-uniform), uniform / clustered / sheet / line clouds with optional far
outliers, fixed radius or per-point radii (r_max / r_min up to 6), symmetric or not, random active pairs, strict or contracted
arithmetic, float or double input, a few steps each with perturbation and occasional resizes.

usage: python tools/fuzz_gpu.py [--minutes 5] [--seed 1] [--devices 3]        (needs a GPU; test infrastructure, not part of the product)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import treensearch_amd as T   # noqa: E402
from oracle import oracle as O   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=5.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--devices", type=int, default=0, help="> 1: A is a multi-device context of that many engines on GPU 0 (tnsx_options.n_devices)")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
orc = O.Oracle()


def cloud(n, kind, scale):
    if n == 0:
        return np.zeros((0, 3), np.float32)
    if kind == "uniform":
        p = rng.random((n, 3))
    elif kind == "clustered":
        k = int(rng.integers(1, 6))
        centres = rng.random((k, 3))
        p = centres[rng.integers(0, k, n)] + rng.normal(0.0, 0.03, (n, 3))
    elif kind == "sheet":
        p = rng.random((n, 3)); p[:, 2] = 0.5 + 0.01 * rng.random(n)
    else:   # line
        t = rng.random(n)
        p = np.stack([t, 0.3 + 0.4 * t, 0.7 - 0.2 * t], axis=1) + rng.normal(0.0, 0.002, (n, 3))
    return (p * scale).astype(np.float32)


def sorted_csr(csr):
    off, idx = csr
    off = np.asarray(off, np.int64); idx = np.asarray(idx, np.int64)
    lid = np.repeat(np.arange(len(off) - 1), np.diff(off))
    order = np.lexsort((idx, lid))
    return off, idx[order]


def same(a, b):
    (oa, ia), (ob, ib) = sorted_csr(a), sorted_csr(b)
    return len(oa) == len(ob) and np.array_equal(oa, ob) and np.array_equal(ia, ib)


t_end = time.time() + 60.0 * args.minutes
n_scenes = n_runs = n_lists = n_oracle = 0
while time.time() < t_end:
    n_sets = int(rng.integers(1, 4))
    scale = float(10.0 ** rng.uniform(-1.0, 2.0))
    sizes = [int(10.0 ** rng.uniform(0.0, 5.3 if rng.random() < 0.1 else 4.5)) if rng.random() > 0.08 else 0 for _ in range(n_sets)]
    kinds = [str(rng.choice(["uniform", "clustered", "sheet", "line"])) for _ in range(n_sets)]
    variable = bool(rng.random() < 0.4)
    symmetric = bool(rng.random() < 0.6)
    arith = int(rng.integers(0, 2))
    use_double = bool(rng.random() < 0.2)
    outliers = bool(rng.random() < 0.25)
    n_tot = max(sum(sizes), 1)
    r0 = scale * 0.6 * (30.0 / n_tot) ** (1.0 / 3.0) * float(rng.uniform(0.3, 1.1))
    r0 = min(r0, 0.2 * scale)
    ratio = float(rng.choice([1.5, 2.5, 6.0]))
    pairs = [(i, j) for i in range(n_sets) for j in range(n_sets) if rng.random() < 0.6] or [(0, 0)]
    # bucket_build_min_points: 1 = the two-pass bucket build for every set whose key fits (round 3), -1 = never, 0 = the default threshold
    opts_a = dict(arith=arith, sorted_lists=bool(rng.random() < 0.3), temporal_reuse=bool(rng.random() < 0.8),
                  bucket_build_min_points=int(rng.choice([1, 1, 0, -1])), query_formulation=int(rng.random() < 0.5))
    if args.devices <= 1 and rng.random() < 0.3:
        # round 4: the sparse grid (lists of occupied cells + block index instead of a dense table) for every scene whose grid has more than a few thousand cells
        opts_a.update(max_dense_cells=int(rng.choice([512, 4096, 1 << 16])), sparse_grid=1)
    A = T.TreeNSearch(**opts_a, devices=[0] * args.devices) if args.devices > 1 else T.TreeNSearch(**opts_a)
    B = T.TreeNSearch(arith=arith, exact_layout=True, temporal_reuse=False)
    pts, rad = [], []
    for s in range(n_sets):
        p = cloud(sizes[s], kinds[s], scale)
        if outliers and len(p) > 3:
            k = int(rng.integers(1, 4))
            p[-k:] = (rng.random((k, 3)) * 2.0 - 1.0) * scale * float(rng.uniform(20.0, 200.0))
        r = (r0 * (1.0 + (ratio - 1.0) * rng.random(len(p)))).astype(np.float32) if variable else None
        pts.append(p.astype(np.float64) if use_double else p)
        rad.append(None if r is None else (r.astype(np.float64) if use_double else r))
    if not variable:
        A.set_search_radius(r0); B.set_search_radius(r0)
    desc = f"sets {sizes} {kinds} scale {scale:.3g} r0 {r0:.3g} variable {variable} ratio {ratio} sym {symmetric} arith {arith} double {use_double} outliers {outliers} pairs {pairs} opts {opts_a}"
    try:
        for s in range(n_sets):
            A.add_point_set(pts[s], rad[s]); B.add_point_set(pts[s], rad[s])
        for (i, j) in pairs:
            A.set_active_search(i, j, True); B.set_active_search(i, j, True)
        A.set_symmetric_search(symmetric); B.set_symmetric_search(symmetric)
        for step in range(int(rng.integers(1, 4))):
            try:
                A.run()
            except T.TnsxError as e:
                # the reference's own limits (2^15 cells per axis, ...): B must refuse as well
                try:
                    B.run()
                    raise AssertionError(f"A refused ({e}) but B ran")
                except T.TnsxError:
                    break
            B.run()
            n_runs += 1
            for (i, j) in pairs:
                a, b = A.neighbor_csr(i, j), B.neighbor_csr(i, j)
                assert same(a, b), f"pool mode differs from exact layout: pair {i}->{j} step {step}"
                n_lists += len(a[0]) - 1
                if len(pts[i]) * max(len(pts[j]), 1) <= 1e8 and len(pts[i]) > 0:
                    f32 = lambda x: np.ascontiguousarray(x, np.float32)
                    if variable:
                        ref = orc.pair_search(f32(pts[i]), f32(pts[j]), ra=f32(rad[i]), rb=f32(rad[j]), symmetric=symmetric, same_set=(i == j), mode=arith, use_grid=False)
                    else:
                        ref = orc.pair_search(f32(pts[i]), f32(pts[j]), radius=r0, same_set=(i == j), mode=arith, use_grid=False)
                    assert same(a, ref), f"engine differs from the all-pairs oracle: pair {i}->{j} step {step}"
                    n_oracle += 1
            # next step: perturb, sometimes resize
            for s in range(n_sets):
                if len(pts[s]) == 0:
                    continue
                if rng.random() < 0.3:
                    keep = int(len(pts[s]) * rng.uniform(0.3, 1.0))
                    pts[s] = np.ascontiguousarray(pts[s][:keep]); rad[s] = None if rad[s] is None else np.ascontiguousarray(rad[s][:keep])
                else:
                    pts[s] = pts[s] + (rng.random(pts[s].shape) - 0.5).astype(pts[s].dtype) * pts[s].dtype.type(0.3 * r0)
                A.resize_point_set(s, pts[s], rad[s]); B.resize_point_set(s, pts[s], rad[s])
    except AssertionError as e:
        print("FAILED:", e); print("  scene:", desc); sys.exit(1)
    n_scenes += 1
    del A, B
    if n_scenes % 10 == 0:
        print(f"  {n_scenes} scenes, {n_runs} runs, {n_lists} lists, {n_oracle} oracle searches", flush=True)
print(f"fuzz ok: {n_scenes} scenes, {n_runs} runs, {n_lists} lists compared between pool mode and exact layout, {n_oracle} pair searches against the all-pairs oracle")
