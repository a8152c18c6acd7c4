#!/usr/bin/env python3
"""Does the C2 query time depend on WHERE the engine's buffers land?  N engines created one after the other in one process (kept alive, or each
destroyed before the next), the same build, the same points; per engine: median query / build stage time and the device addresses of the record
pool and the offsets (profiles/r2_pool_regions_ab.txt "order effects": +-5 % between engines that differ in nothing else).
usage: placement_probe.py [--engines 6] [--keep] [--pad MB ...]   --pad: allocate and hold that many MB of device memory before every engine"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd as T
from treensearch_amd import datagen as D
ap = argparse.ArgumentParser(); ap.add_argument("--engines", type=int, default=6); ap.add_argument("--keep", action="store_true")
ap.add_argument("--pad", type=int, nargs="*", default=[]); ap.add_argument("--steps", type=int, default=12); ap.add_argument("--lib", default=None)
args = ap.parse_args()
if args.lib:
    import treensearch_amd.api as A
    A._lib = None; A.LIB_PATH = os.path.abspath(args.lib)
n = 10_000_000
base = torch.from_numpy(D.uniform_cloud(n, 12345)).cuda()
radius = D.radius_for_neighbors(n)
g = torch.Generator(device="cuda").manual_seed(1)
dlt = (torch.rand(base.shape, generator=g, device="cuda", dtype=torch.float32) - 0.5) * (2.0 * 0.1 * float(radius) / 3.0 ** 0.5)
copies = [base + dlt, base - dlt]
held, pads = [], []
for e in range(args.engines):
    if args.pad:
        pads.append(torch.empty(args.pad[e % len(args.pad)] << 20, dtype=torch.uint8, device="cuda"))
    ns = T.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=True)
    ns.set_search_radius(radius); ns.add_point_set(copies[0]); ns.set_active_search(0, 0, True)
    fill, sort = [], []
    for k in range(4 + args.steps):
        ns.resize_point_set(0, copies[k % 2]); ns.run()
        if k >= 4:
            st = ns.get_stats(); fill.append(st["ms_fill"]); sort.append(st["ms_sort"])
    v = ns.pair_view(0, 0)
    print(f"engine {e}: query med {np.median(fill):.4f} min {np.min(fill):.4f} | build med {np.median(sort):.4f} | records @ {v.records_device:#x} "
          f"(mod 2M {v.records_device % (2 << 20):#x}, mod 1G {v.records_device % (1 << 30):#x}) offsets @ {v.offsets_device:#x} n_records {v.n_records}", flush=True)
    if args.keep: held.append(ns)
    else: del ns
