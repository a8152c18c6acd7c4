#!/usr/bin/env python3
"""Sparse grid against the coarsened dense grid on clouds that are sparse everywhere (10 M points): a filament (1-D) and a sheet (2-D) through the whole box.
usage: sparse_probe.py [filament|sheet] [sparse|coarse]   (one engine per process: meant to run under rocprofv3 --kernel-trace --stats)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd as T
kind = sys.argv[1] if len(sys.argv) > 1 else "filament"
mode = sys.argv[2] if len(sys.argv) > 2 else "sparse"
n = 10_000_000
rng = np.random.default_rng(3)
if kind == "filament":
    t = np.sort(rng.random(n)); ang = 2.0 * np.pi * 420.0 * t; r = np.float32(0.00075)
    pts = np.stack([0.5 + 0.45 * np.cos(ang), 0.5 + 0.45 * np.sin(ang), 0.02 + 0.96 * t], axis=1) + (rng.random((n, 3)) - 0.5) * (0.6 * float(r))
else:
    # a sphere shell of radius 0.48: ~30 neighbours at r = 0.00166 -> a box of ~2 x 10^8 cells... thinner radius for more cells
    v = rng.standard_normal((n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True); r = np.float32(0.0012)
    pts = 0.5 + 0.48 * v + (rng.random((n, 3)) - 0.5) * (0.5 * float(r))
pts = np.ascontiguousarray(pts.astype(np.float32))
d = torch.from_numpy(pts).cuda()
g = torch.Generator(device="cuda").manual_seed(1)
dl = (torch.rand(d.shape, generator=g, device="cuda") - 0.5) * (0.1 * float(r))
copies = [d + dl, d - dl]
ns = T.TreeNSearch(sparse_grid=(0 if mode == "sparse" else -1), collect_stage_times=True)
ns.set_search_radius(r); ns.add_point_set(copies[0]); ns.set_active_search(0, 0, True)
for k in range(4):
    ns.resize_point_set(0, copies[k % 2]); ns.run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(6):
    ns.resize_point_set(0, copies[k % 2]); ns.run()
torch.cuda.synchronize()
st = ns.get_stats()
print(f"{kind} {mode}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms per run (moving points) | grid {st['grid_dims']} sparse {st['grid_sparse']} cell/r {st['grid_cell_size'] / float(r):.3f} "
      f"occupied {st['n_occupied_cells']} of {st['n_grid_cells']} | neighbours/point {st['n_neighbors'] / n:.1f} | build {st['ms_sort']:.2f} query {st['ms_fill']:.2f} ms, passes {st['radix_passes']}")
