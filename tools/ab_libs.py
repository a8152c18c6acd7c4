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
ap.add_argument("--recreate", type=int, default=0, help="R > 0: every build gets R engines, created, timed and destroyed one after the other in rotating order -- where an "
                "engine's buffers land physically moves the C2 query by +-5 % (tools/placement_probe.py), more than most kernel changes; the mean over R placements is what compares")
ap.add_argument("--zsort", action="store_true", help="c2: hand the points over in z-order (prepare_zsort + apply_zsort once)")
ap.add_argument("--move", action="store_true", help="c2: the points move between runs like in bench.py (two copies, +-0.058 r per coordinate)")
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
if args.workload == "c2" and args.zsort:
    tmp = A.TreeNSearch(); tmp.set_search_radius(radius); tmp.add_point_set(sets[0][0]); tmp.prepare_zsort(); tmp.apply_zsort(0, sets[0][0], 3); torch.cuda.synchronize(); del tmp
copies = None
if args.workload == "c2" and args.move:
    g = torch.Generator(device="cuda").manual_seed(1)
    dlt = (torch.rand(sets[0][0].shape, generator=g, device="cuda", dtype=torch.float32) - 0.5) * (2.0 * 0.1 * float(radius) / 3.0 ** 0.5)
    copies = [sets[0][0] + dlt, sets[0][0] - dlt]
def make(path):
    A._lib = None; A.LIB_PATH = os.path.abspath(path)
    ns = A.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=True)
    if radius is not None: ns.set_search_radius(radius)
    for (p, r) in sets: ns.add_point_set(p, r)
    for (i, j) in pairs: ns.set_active_search(i, j, True)
    return ns
if args.recreate > 0:
    res = {p: dict(fill=[], sort=[], total=[]) for p in args.libs}
    for r in range(args.recreate):
        order = args.libs[r % len(args.libs):] + args.libs[:r % len(args.libs)]
        for path in order:
            ns = make(path)
            f, so, t = [], [], []
            for it in range(4 + args.steps):
                if copies is not None: ns.resize_point_set(0, copies[it % 2])
                ns.run()
                if it >= 4:
                    st = ns.get_stats(); f.append(st["ms_fill"]); so.append(st["ms_sort"]); t.append(st["ms_total"])
            res[path]["fill"].append(np.median(f)); res[path]["sort"].append(np.median(so)); res[path]["total"].append(np.median(t))
            del ns
    for path in args.libs:
        a = res[path]
        print(f"{os.path.basename(path):28s} over {args.recreate} placements: fill mean {np.mean(a['fill']):.4f} min {np.min(a['fill']):.4f} max {np.max(a['fill']):.4f} | "
              f"sort mean {np.mean(a['sort']):.4f} | total mean {np.mean(a['total']):.4f} min {np.min(a['total']):.4f}")
    sys.exit(0)
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
        for it in range(args.steps):
            if copies is not None: ns.resize_point_set(0, copies[it % 2])
            t_w = time.perf_counter(); ns.run(); acc[k]["wall"].append((time.perf_counter() - t_w) * 1e3); st = ns.get_stats()
            acc[k]["fill"].append(st["ms_fill"]); acc[k]["sort"].append(st["ms_sort"]); acc[k]["total"].append(st["ms_total"]); acc[k]["retries"] += st.get("pool_retries", 0)
for path, a in zip(args.libs, acc):
    print(f"{os.path.basename(path):28s} fill med {np.median(a['fill']):.4f} min {np.min(a['fill']):.4f} | sort med {np.median(a['sort']):.4f} | total med {np.median(a['total']):.4f} min {np.min(a['total']):.4f} | wall med {np.median(a['wall']):.4f} | pool retries {a['retries']}")
