#!/usr/bin/env python3
"""Where does the per-step time of SlabSearch go on one rank without peers? (python / torch overhead of the slab layer)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd as T
from treensearch_amd import datagen as D
from treensearch_amd.multi import SlabSearch
n = 10_000_000
radius = D.radius_for_neighbors(n)
pts_h = D.uniform_cloud(n, 12345)
slab = SlabSearch(0.0, 1.0, float(radius), lambda: T.TreeNSearch(collect_stage_times=True))
d_pts = slab.owned_buffer(n, "cuda"); d_pts.copy_(torch.from_numpy(pts_h))
gids = torch.arange(n, dtype=torch.int64, device="cuda")
for _ in range(3): slab.step(d_pts, gids)
acc = {"exchange": 0.0, "bookkeeping": 0.0, "run": 0.0, "device": 0.0}
steps = 20
e = slab.engine
for _ in range(steps):
    t0 = time.perf_counter(); g = slab.ex.exchange(d_pts, gids); t1 = time.perf_counter()
    e.resize_point_set(slab._set, slab._buf[:n]); t2 = time.perf_counter()
    e.run(); t3 = time.perf_counter()
    acc["exchange"] += t1 - t0; acc["bookkeeping"] += t2 - t1; acc["run"] += t3 - t2; acc["device"] += e.get_stats()["ms_total"] / 1e3
print({k: round(v / steps * 1e3, 4) for k, v in acc.items()})
st = e.get_stats(); print("slab stages", {k[3:]: round(st[k], 4) for k in st if k.startswith("ms_")})
ns = T.TreeNSearch(collect_stage_times=True); ns.set_search_radius(radius); ns.add_point_set(torch.from_numpy(pts_h).cuda()); ns.set_active_search(0, 0, True)
for _ in range(3): ns.run()
t0 = time.perf_counter()
for _ in range(steps): ns.run()
st = ns.get_stats(); print("plain stages", {k[3:]: round(st[k], 4) for k in st if k.startswith("ms_")})
print("plain run():", round((time.perf_counter() - t0) / steps * 1e3, 4), "ms, device", round(ns.get_stats()["ms_total"], 4))
