#!/usr/bin/env python3
"""C2 (10 M uniform points, z-ordered, moving) through ONE build of the library, for profiler passes of a variant:  tools/c2_loop.py <libtnsx.so> [steps] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd.api as A
from treensearch_amd import datagen as D
lib = os.path.abspath(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8; n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
A._lib, A.LIB_PATH = None, lib
base = D.uniform_cloud_torch(n, 12345); radius = D.radius_for_neighbors(n)
tmp = A.TreeNSearch(); tmp.set_search_radius(radius); tmp.add_point_set(base); tmp.prepare_zsort(); tmp.apply_zsort(0, base, 3); torch.cuda.synchronize(); del tmp
g = torch.Generator(device="cuda").manual_seed(1)
dlt = (torch.rand(base.shape, generator=g, device="cuda", dtype=torch.float32) - 0.5) * (2.0 * 0.1 * float(radius) / 3.0 ** 0.5)
copies = [base + dlt, base - dlt]
ns = A.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=True)
ns.set_search_radius(radius); ns.add_point_set(copies[0]); ns.set_active_search(0, 0, True)
fill = []
for k in range(steps):
    ns.resize_point_set(0, copies[k % 2]); ns.run(); fill.append(ns.get_stats()["ms_fill"])
print(f"{os.path.basename(lib)}: fill ms (steady) med {np.median(fill[3:]):.4f} min {np.min(fill[3:]):.4f} | neighbours {ns.get_stats()['n_neighbors']}")
