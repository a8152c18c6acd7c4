#!/usr/bin/env python3
"""prepare_zsort of several builds of libtnsx.so in ONE process, interleaved (dam-break cloud kept in z-order, per-point radii; the cell-level order after a run()).
usage: zsort_ab.py lib_a.so lib_b.so ... [--points N] [--rounds R]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd.api as A
from treensearch_amd import datagen as D
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--points", type=int, default=10_000_000); ap.add_argument("--rounds", type=int, default=12)
args = ap.parse_args()
p, rad, r0 = D.dam_break_cloud(args.points, 1)
d_p, d_r = torch.from_numpy(p).cuda(), torch.from_numpy(rad).cuda()
def make(path):
    A._lib = None; A.LIB_PATH = os.path.abspath(path)
    ns = A.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=False)
    ns.add_point_set(d_p, d_r); ns.set_active_search(0, 0, True); ns.set_symmetric_search(True); ns._libpath = path
    return ns
first = make(args.libs[0]); first.run(); first.prepare_zsort(); first.apply_zsort(0, d_p, 3); first.apply_zsort(0, d_r, 1); torch.cuda.synchronize(); del first
engines = [make(l) for l in args.libs]
ref = None
for ns in engines:
    ns.run(); ns.prepare_zsort()
    o = ns.get_zsort_order(0) if hasattr(ns, "get_zsort_order") else None
    if o is not None:
        o = torch.as_tensor(o).cpu().numpy()
        if not np.array_equal(np.sort(o), np.arange(args.points)): print(f"!!! {os.path.basename(ns._libpath)}: the order is NOT a permutation ({args.points - len(np.unique(o))} duplicates)", flush=True)
t = {l: [] for l in args.libs}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for r in range(args.rounds):
    order = list(range(len(engines))); order = order[r % len(order):] + order[:r % len(order)]
    for k in order:
        ns = engines[k]
        ns.run()
        ev[0].record(); ns.prepare_zsort(); ev[1].record(); ev[1].synchronize()
        if r >= 2: t[args.libs[k]].append(ev[0].elapsed_time(ev[1]))
for l in args.libs:
    print(f"{os.path.basename(l):28s} prepare_zsort mean {np.mean(t[l]):.4f} ms  min {np.min(t[l]):.4f}  max {np.max(t[l]):.4f}  ({args.points} points, {len(t[l])} rounds)")
