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
ap.add_argument("--check", action="store_true", help="compare the neighbour lists of every build with the first one's (order-independent digest, oracle/tns_oracle.c: test infrastructure)")
ap.add_argument("--workload", choices=["c2", "c3", "c4"], default="c2", help="c2 uniform fixed radius; c3 two sets 0->0,0->1; c4 dam break, per-point radii, symmetric")
args = ap.parse_args()
n = args.points
if args.workload == "c2":
    sets = [(torch.from_numpy(D.uniform_cloud(n, 12345)).cuda(), None)]; radius = D.radius_for_neighbors(n); pairs = [(0, 0)]
elif args.workload == "c3":
    f, b, radius = D.two_set_cloud(int(0.8 * n), n - int(0.8 * n))
    sets = [(torch.from_numpy(f).cuda(), None), (torch.from_numpy(b).cuda(), None)]; pairs = [(0, 0), (0, 1)]
else:
    p, rad, _ = D.dam_break_cloud(n)
    sets = [(torch.from_numpy(p).cuda(), torch.from_numpy(rad).cuda())]; radius = None; pairs = [(0, 0)]
engines = []
for path in args.libs:
    A._lib = None; A.LIB_PATH = os.path.abspath(path)
    ns = A.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=True)
    if radius is not None: ns.set_search_radius(radius)
    for (p, r) in sets: ns.add_point_set(p, r)
    for (i, j) in pairs: ns.set_active_search(i, j, True)
    for _ in range(3): ns.run()
    engines.append(ns)
if args.check:
    from oracle import oracle as O
    orc = O.Oracle()
    digs = []
    for path, ns in zip(args.libs, engines):
        d = []
        for (i, j) in pairs:
            offs, idx = ns.neighbor_csr(i, j, sort_each=False)
            d.append((int(offs[-1]),) + orc.digest(offs, idx, already_sorted=False))
        digs.append(d)
        print(f"{os.path.basename(path):28s} lists {'== first build' if d == digs[0] else '!!! DIFFER from the first build !!!'} {d}", flush=True)
import time
acc = [dict(fill=[], sort=[], total=[], wall=[], retries=0) for _ in engines]
for r in range(args.rounds):
    for k, ns in enumerate(engines):
        for _ in range(args.steps):
            t_w = time.perf_counter(); ns.run(); acc[k]["wall"].append((time.perf_counter() - t_w) * 1e3); st = ns.get_stats()
            acc[k]["fill"].append(st["ms_fill"]); acc[k]["sort"].append(st["ms_sort"]); acc[k]["total"].append(st["ms_total"]); acc[k]["retries"] += st.get("pool_retries", 0)
for path, a in zip(args.libs, acc):
    print(f"{os.path.basename(path):28s} fill med {np.median(a['fill']):.4f} min {np.min(a['fill']):.4f} | sort med {np.median(a['sort']):.4f} | total med {np.median(a['total']):.4f} min {np.min(a['total']):.4f} | wall med {np.median(a['wall']):.4f} | pool retries {a['retries']}")
