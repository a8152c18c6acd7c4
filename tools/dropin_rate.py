#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in mode (what a CPU consumer of the C++ shim sees): host arrays in, neighbour lists mirrored
into pinned host memory.  BASELINE configs[1], 10 M uniform points.
usage: dropin_rate.py [n_points] [device list, e.g. 0,1,2,3 -> multi-device mode of the ABI, or - ] [library to load instead of the default one]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import treensearch_amd as T
import treensearch_amd.api as A
from treensearch_amd import datagen as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
devices = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
if len(sys.argv) > 3:
    A._lib = None; A.LIB_PATH = os.path.abspath(sys.argv[3])
pts = D.uniform_cloud(n, 12345)
ns = T.TreeNSearch(mirror_to_host=True, collect_stage_times=True, devices=devices)
ns.set_search_radius(D.radius_for_neighbors(n)); ns.add_point_set(pts); ns.set_active_search(0, 0, True)
for _ in range(2): ns.run()
t0 = time.perf_counter()
steps = 5
for k in range(steps):
    pts[k::97, 1] += np.float32(1e-6)        # (the points change between runs, like in the simulation this mode serves)
    ns.run()
ms = (time.perf_counter() - t0) / steps * 1e3
st = ns.get_stats()
v = ns.pair_view(0, 0)
if devices and len(devices) > 1:
    print(f"drop-in mode on engines {devices}: {ms:.2f} ms per run() = {n / ms / 1e3:.1f} Mpoints/s; {st['n_devices_used']} slabs, {v.n_records * 4 / 1e9:.2f} GB of records gathered "
          f"into one pinned buffer, slowest engine {st['ms_total']:.2f} ms of device work + upload")
else:
    print(f"drop-in mode ({os.path.basename(A.LIB_PATH)}): {ms:.2f} ms per run() = {n / ms / 1e3:.1f} Mpoints/s; upload {st['ms_upload']:.2f} ms, device work {st['ms_total'] - st['ms_upload'] - st['ms_mirror']:.2f} ms, "
          f"mirror of {((v.n_neighbors + v.n_points) * 4 + v.n_points * 8) / 1e9:.2f} GB (gap-free records + offsets; the pool on the device: {v.n_records * 4 / 1e9:.2f} GB) {st['ms_mirror']:.2f} ms")
