#!/usr/bin/env python3
"""Where the z-sort of a C4-shaped step goes: prepare_zsort, apply_zsort(xyz), apply_zsort(radii) timed one by one (a synchronisation between them) on a dam-break
cloud that is kept in z-order like the bench's c4 workload.  usage: zsort_probe.py [points] [steps]   (under rocprofv3 --kernel-trace --stats for the kernels)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd as T
from treensearch_amd import datagen as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
p, rad, r0 = D.dam_break_cloud(n, 1)
d_p, d_r = torch.from_numpy(p).cuda(), torch.from_numpy(rad).cuda()
g = torch.Generator(device="cuda").manual_seed(3)
d_delta = (torch.rand(d_p.shape, generator=g, device="cuda", dtype=torch.float32) - 0.5) * (2.0 * 0.1 * float(r0) / 3.0 ** 0.5)
ns = T.TreeNSearch(collect_stage_times=False)
ns.add_point_set(d_p, d_r); ns.set_active_search(0, 0, True); ns.set_symmetric_search(True)
t = {"prepare": [], "apply_xyz": [], "apply_r": [], "run": []}
def timed(name, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); t[name].append((time.perf_counter() - t0) * 1e3)
for k in range(3 + steps):
    d_p.add_(d_delta, alpha=1.0 if k % 2 == 0 else -1.0)
    timed("prepare", ns.prepare_zsort)
    timed("apply_xyz", lambda: ns.apply_zsort(0, d_p, 3))
    timed("apply_r", lambda: ns.apply_zsort(0, d_r, 1))
    timed("run", ns.run)
m = {k: float(np.mean(v[3:])) for k, v in t.items()}
print(f"{n} points, {steps} steps: prepare_zsort {m['prepare']:.3f} ms | apply_zsort(xyz) {m['apply_xyz']:.3f} | apply_zsort(radii) {m['apply_r']:.3f} | "
      f"z-sort total {m['prepare'] + m['apply_xyz'] + m['apply_r']:.3f} | run {m['run']:.3f}")
