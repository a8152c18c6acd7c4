#!/usr/bin/env python3
"""A/B timing of several builds of libtnsx.so inside ONE process, interleaved round by round (the spread between processes,
boxes and clock states is larger than most kernel tweaks).  usage: ab_libs.py lib_a.so lib_b.so ... [--rounds R] [--steps K]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd.api as A
from treensearch_amd import datagen as D
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--rounds", type=int, default=6); ap.add_argument("--steps", type=int, default=15)
ap.add_argument("--points", type=int, default=10_000_000)
args = ap.parse_args()
n = args.points
pts = torch.from_numpy(D.uniform_cloud(n, 12345)).cuda()
engines = []
for path in args.libs:
    A._lib = None; A.LIB_PATH = os.path.abspath(path)
    ns = A.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=True)
    ns.set_search_radius(D.radius_for_neighbors(n)); ns.add_point_set(pts); ns.set_active_search(0, 0, True)
    for _ in range(3): ns.run()
    engines.append(ns)
acc = [dict(fill=[], sort=[], total=[]) for _ in engines]
for r in range(args.rounds):
    for k, ns in enumerate(engines):
        for _ in range(args.steps):
            ns.run(); st = ns.get_stats()
            acc[k]["fill"].append(st["ms_fill"]); acc[k]["sort"].append(st["ms_sort"]); acc[k]["total"].append(st["ms_total"])
for path, a in zip(args.libs, acc):
    print(f"{os.path.basename(path):28s} fill med {np.median(a['fill']):.4f} min {np.min(a['fill']):.4f} | sort med {np.median(a['sort']):.4f} | total med {np.median(a['total']):.4f} min {np.min(a['total']):.4f}")
