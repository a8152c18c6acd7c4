#!/usr/bin/env python3
"""C2 workload through one query formulation: stage times + how much the group kernel passed on (tools/group_probe.py [formulation] [n])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import treensearch_amd as T
import treensearch_amd.api as A
# the group formulation is not part of the product library: load the variant that carries it (tools/build_group_variant.sh)
VARIANT = os.path.join(ROOT, "ab_libs", "libtnsx_group.so")
if os.path.exists(VARIANT):
    A._lib, A.LIB_PATH = None, VARIANT
from treensearch_amd import datagen as D
form = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
pts = torch.from_numpy(D.uniform_cloud(n, 12345)).cuda()
ns = T.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=True, query_formulation=form)
ns.set_search_radius(D.radius_for_neighbors(n)); ns.add_point_set(pts); ns.set_active_search(0, 0, True)
fill, tot = [], []
for i in range(steps):
    ns.run(); st = ns.get_stats(); fill.append(st["ms_fill"]); tot.append(st["ms_total"])
f = np.array(fill[3:]); t = np.array(tot[3:])
print("formulation %d n %d: fill min %.3f med %.3f | total med %.3f | neighbours %d | group pairs %d, cells passed on %d (of %d occupied)"
      % (form, n, f.min(), np.median(f), np.median(t), st["n_neighbors"], st["n_group_pairs"], st["n_group_passed_cells"], st["n_occupied_cells"]))
