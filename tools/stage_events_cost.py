#!/usr/bin/env python3
"""What do the stage-time events cost?  C2, points moving between two copies, wall time per run() with collect_stage_times on / off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd as T
from treensearch_amd import datagen as D
n = 10_000_000
r = D.radius_for_neighbors(n)
base = torch.from_numpy(D.uniform_cloud(n, 12345)).cuda()
d = (torch.rand(base.shape, device="cuda") - 0.5) * (0.2 * float(r) / 3 ** 0.5)
copies = [base + d, base - d]
for rep in range(2):
    for collect in (True, False):
        ns = T.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=collect)
        ns.set_search_radius(r); ns.add_point_set(copies[0]); ns.set_active_search(0, 0, True)
        for k in range(4):
            ns.resize_point_set(0, copies[k % 2]); ns.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(30):
            ns.resize_point_set(0, copies[k % 2]); ns.run()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 30 * 1e3
        print("collect_stage_times %-5s: %.4f ms per run (device total of the last run %.4f)" % (collect, t, ns.get_stats()["ms_total"]))
        del ns
