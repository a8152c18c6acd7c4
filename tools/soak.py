#!/usr/bin/env python3
"""Soak test: the same inputs through many run()s must give the same neighbour SETS every time (record order and pool layout
may differ).  Two order-independent digests per run: sum_i w_i * |N(i)| and sum_i w_i * sum_{j in N(i)} j.  Catches races in the
emission / allocation / cull paths that a single parity run can miss."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd as T
from treensearch_amd import datagen as D

def digests(ns, n, w):
    offs, recs = ns.neighbor_records(0, 0)
    offs = offs.astype(np.int64)
    cnt = recs[offs].astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(recs.astype(np.int64))])
    sums = csum[offs + 1 + cnt] - csum[offs + 1]
    with np.errstate(over="ignore"):
        return int(np.sum(w * cnt.astype(np.uint64), dtype=np.uint64)), int(np.sum(w * sums.astype(np.uint64), dtype=np.uint64)), int(cnt.sum())

def soak(name, make, runs):
    ns, n = make()
    w = np.random.default_rng(1).integers(1, 1 << 40, n, dtype=np.uint64)
    ns.run()
    ref = digests(ns, n, w)
    t0 = time.time(); bad = 0
    for r in range(runs):
        ns.run()
        ns._views = {}
        d = digests(ns, n, w)
        if d != ref:
            bad += 1
            print(f"  {name}: run {r} differs: {d} vs {ref}")
    print(f"{name}: {runs} runs, {bad} differing, neighbours {ref[2]}, {time.time() - t0:.0f} s")
    return bad

def uniform():
    n = 2_000_000
    ns = T.TreeNSearch(); ns.set_search_radius(D.radius_for_neighbors(n)); ns.add_point_set(torch.from_numpy(D.uniform_cloud(n, 5)).cuda()); ns.set_active_search(0, 0, True)
    return ns, n
def dam():
    n = 2_000_000
    p, r, _ = D.dam_break_cloud(n)
    ns = T.TreeNSearch(); ns.add_point_set(torch.from_numpy(p).cuda(), torch.from_numpy(r).cuda()); ns.set_active_search(0, 0, True); ns.set_symmetric_search(True)
    return ns, n
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sys.exit(1 if soak("uniform 2M", uniform, runs) + soak("dam break 2M (per-point radii, symmetric, cull path)", dam, runs) else 0)
