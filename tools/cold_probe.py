#!/usr/bin/env python3
"""Where the time of a COLD run() goes (first run of an engine: allocations, bounds, first grid, the count-only pass that sizes the pool, the sized pass).
usage: tools/cold_probe.py [n_points] [engines]   -- engine 0 also pays the process's one-off costs (code object load), engines 1.. only their own."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import treensearch_amd as T
from treensearch_amd import datagen as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
engines = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pts = D.uniform_cloud_torch(n, 12345)
r = D.radius_for_neighbors(n)
torch.cuda.synchronize()
for k in range(engines):
    t0 = time.perf_counter()
    ns = T.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=True)
    ns.set_search_radius(r); ns.add_point_set(pts); ns.set_active_search(0, 0, True)
    t1 = time.perf_counter()
    ns.run(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = ns.get_stats()
    ns.run(); torch.cuda.synchronize()
    t3 = time.perf_counter()
    st2 = ns.get_stats()
    print(f"engine {k}: create {1e3 * (t1 - t0):.2f} ms | cold run {1e3 * (t2 - t1):.2f} ms wall (stages of its last attempt: total {st['ms_total']:.2f} bounds {st['ms_bounds']:.2f} "
          f"sort {st['ms_sort']:.2f} cells {st['ms_cells']:.2f} fill {st['ms_fill']:.2f}; cold passes {st['cold_passes']}, sampled {st['sampled_passes']}, pool retries {st['pool_retries']}) | "
          f"second run {1e3 * (t3 - t2):.2f} ms wall (total {st2['ms_total']:.2f} fill {st2['ms_fill']:.2f}) | neighbours {st['n_neighbors']}")
    t4 = time.perf_counter()
    del ns
    torch.cuda.synchronize()
    print(f"          destroy {1e3 * (time.perf_counter() - t4):.2f} ms")
