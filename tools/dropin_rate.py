#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in mode (what a CPU consumer of the C++ shim sees): host arrays in, neighbour lists mirrored
into pinned host memory.  BASELINE configs[1], 10 M uniform points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import treensearch_amd as T
from treensearch_amd import datagen as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts = D.uniform_cloud(n, 12345)
ns = T.TreeNSearch(mirror_to_host=True, collect_stage_times=True)
ns.set_search_radius(D.radius_for_neighbors(n)); ns.add_point_set(pts); ns.set_active_search(0, 0, True)
for _ in range(2): ns.run()
t0 = time.perf_counter()
steps = 5
for _ in range(steps): ns.run()
ms = (time.perf_counter() - t0) / steps * 1e3
st = ns.get_stats()
v = ns.pair_view(0, 0)
print(f"drop-in mode: {ms:.2f} ms per run() = {n / ms / 1e3:.1f} Mpoints/s; upload {st['ms_upload']:.2f} ms, device work {st['ms_total'] - st['ms_upload'] - st['ms_mirror']:.2f} ms, "
      f"mirror of {v.n_records * 4 / 1e9:.2f} GB {st['ms_mirror']:.2f} ms")
