#!/usr/bin/env python3
"""Per-step stage times of the C2 workload (is the run-to-run spread of bench.py a drift inside a process or a property of the process?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd as T
from treensearch_amd import datagen as D
n = 10_000_000
pts = torch.from_numpy(D.uniform_cloud(n, 12345)).cuda()
ns = T.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=True)
ns.set_search_radius(D.radius_for_neighbors(n)); ns.add_point_set(pts); ns.set_active_search(0, 0, True)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
fill, tot = [], []
for i in range(steps):
    ns.run(); st = ns.get_stats(); fill.append(st["ms_fill"]); tot.append(st["ms_total"])
f = np.array(fill[3:]); t = np.array(tot[3:])
print("fill  min %.3f med %.3f max %.3f | first 10:" % (f.min(), np.median(f), f.max()), np.round(f[:10], 3), "last 5:", np.round(f[-5:], 3))
print("total min %.3f med %.3f max %.3f" % (t.min(), np.median(t), t.max()))
