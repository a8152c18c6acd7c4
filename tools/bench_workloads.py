#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md: BASELINE.json configs[2] (two sets, asymmetric searches) and configs[3]-shaped
(dam break, per-point radii, symmetric, zsort every step) on one GPU.  Prints one JSON line per workload."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import treensearch_amd as T
from treensearch_amd import datagen as D

def timed(fn, steps, warmup):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--c4-points", type=int, default=50_000_000)
args = ap.parse_args()
stream = torch.cuda.current_stream().cuda_stream

# ---- C3: 8 M fluid + 2 M boundary, searches 0->0 and 0->1
f, b, r = D.two_set_cloud(8_000_000, 2_000_000)
ns = T.TreeNSearch(stream=stream, collect_stage_times=True)
ns.set_search_radius(r)
df, db = torch.from_numpy(f).cuda(), torch.from_numpy(b).cuda()
ns.add_point_set(df); ns.add_point_set(db)
ns.set_active_search(0, 0, True); ns.set_active_search(0, 1, True)
ms = timed(ns.run, args.steps, args.warmup); st = ns.get_stats()
print(json.dumps({"workload": "C3 two sets 8M+2M, 0->0 and 0->1", "ms_per_run": round(ms, 3), "Mpoints_per_s": round(10e6 / ms / 1e3, 1),
                  "neighbors": st["n_neighbors"], "queries": st["n_queries"], "pool_pairs": st["n_pool_pairs"],
                  "stage_ms": {k[3:]: round(st[k], 3) for k in st if k.startswith("ms_")}}), flush=True)
del ns, df, db

# ---- C4-shaped: dam break, per-point radii, symmetric; every step: zsort (prepare + apply to xyz and radii) + run
n = args.c4_points
p, rad, r0 = D.dam_break_cloud(n)
ns = T.TreeNSearch(stream=stream, collect_stage_times=True)
dp, dr = torch.from_numpy(p).cuda(), torch.from_numpy(rad).cuda()
ns.add_point_set(dp, dr); ns.set_active_search(0, 0, True); ns.set_symmetric_search(True)
def step():
    ns.prepare_zsort(); ns.apply_zsort(0, dp, 3); ns.apply_zsort(0, dr, 1); ns.run()
ms = timed(step, max(args.steps // 2, 3), 2); st = ns.get_stats()
ms_run = timed(ns.run, max(args.steps // 2, 3), 1)
print(json.dumps({"workload": f"C4 dam break {n} pts, per-point radii, symmetric", "ms_per_step_with_zsort": round(ms, 3), "ms_run_only": round(ms_run, 3),
                  "Mpoints_per_s_run_only": round(n / ms_run / 1e3, 1), "neighbors": st["n_neighbors"], "grid": st["grid_dims"], "pool_pairs": st["n_pool_pairs"],
                  "stage_ms": {k[3:]: round(st[k], 3) for k in st if k.startswith("ms_")}}), flush=True)
