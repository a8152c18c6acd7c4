#!/usr/bin/env python3
"""bench.py -- neighbour build + query throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one tnsx run() -- world bounds, cell keys, radix sort, gather, cell table, 27-cell query (count, scan,
fill) -- over a batch of synthetic points that are already resident in HBM; the neighbour lists stay in HBM.
N = 1: BASELINE.json configs[1], 10 M uniform points, fixed radius (~59 neighbours).  N > 1: weak scaling, every rank
owns one unit-cube slab of 10 M points (global cloud = N slabs along x) and exchanges one-radius ghost halos with its
slab neighbours over RCCL every step (treensearch_amd/multi.py); value = all points of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line (metric/value/unit/... + "roofline" for the dominant kernel + "cpu_baseline").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def cpu_baseline(n_points: int, radius: float, seed: int):
    """Times the reference's AVX2 path (oracle/_ref, built from /root/reference in the build container) on this
    box's host cores; falls back to the CPU restatement (oracle/) on a smaller sample when the reference build did
    not travel.  Protocol of BASELINE.md section 4: z-sort first, warm-up runs, median of the timed runs."""
    from oracle import oracle as O
    from treensearch_amd import datagen as D
    cores = os.cpu_count() or 1
    try:
        if O.have_ref():
            pts = D.uniform_cloud(n_points, seed)
            ref = O.RefTreeNSearch(strict=False)
            ref.set_search_radius(radius)
            s = ref.add_point_set(pts)
            ref.set_active_search(s, s, True)
            ref.prepare_zsort()
            ref.apply_zsort(s, pts, 3)
            times = []
            for it in range(3 + 5):
                t0 = time.perf_counter()
                ref.run()
                t1 = time.perf_counter()
                if it >= 3:
                    times.append(t1 - t0)
            t = float(np.median(times))
            return {"value": round(n_points / t / 1e6, 3), "unit": "Mpoints/s", "cores": O.Oracle().num_threads(),
                    "kind": "reference", "host_cpus": cores,
                    "sample": f"tns::TreeNSearch::run() (AVX2 path, reference flags, -march=haswell) on the same {n_points} "
                              f"uniform points, z-sorted first, 3 warm-up + median of 5 runs ({t * 1e3:.1f} ms/run)"}
    except Exception as e:  # pragma: no cover - the reference library is optional on the GPU box
        sys.stderr.write(f"[bench] reference baseline unavailable: {e}\n")
    orc = O.Oracle()
    n_s = min(n_points, 1_000_000)
    pts = D.uniform_cloud(n_s, seed)
    r = D.radius_for_neighbors(n_s)
    t0 = time.perf_counter()
    orc.pair_search(pts, pts, radius=r, same_set=True, mode=O.STRICT)
    t = time.perf_counter() - t0
    return {"value": round(n_s / t / 1e6, 3), "unit": "Mpoints/s", "cores": orc.num_threads(), "kind": "port",
            "host_cpus": cores, "sample": f"oracle/tns_oracle.c grid search on {n_s} uniform points, 1 run ({t:.2f} s)"}


def pmc_traffic(arith: str, pooled: bool):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_latest.json,
    written by tools/prof_gpu.sh on the SAME workload): counters cannot be collected inside a timed run.  Corrections as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KiB) counts 128-byte requests of wide coalesced
    reads as 64 bytes -> doubled; WRITE_SIZE (KiB) is taken as reported (uncalibrated)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not pooled or not os.path.exists(path):
        return None, None
    try:
        prof = json.load(open(path))
        name = f"k_query_pool_fast<{0 if arith == 'strict' else 1}, false, false, true, false>"
        pmc = prof["kernels"][name]["pmc"]
        fetch = 2.0 * pmc["FETCH_SIZE"] * 1024.0
        write = pmc["WRITE_SIZE"] * 1024.0
        return int(fetch + write), {"fetch_bytes": int(fetch), "write_bytes": int(write), "source": "profiles/pmc_latest.json (" + prof.get("source", "?") + ")",
                                    "note": "L2<->fabric bytes (Infinity-Cache hits included); FETCH_SIZE x2 per the gfx950 guide, WRITE_SIZE uncalibrated"}
    except (KeyError, ValueError, OSError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=10_000_000, help="points per GPU")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--arith", choices=["strict", "contracted"], default="strict")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-layout", action="store_true", help="two-pass count/scan/fill result layout instead of the single pass")
    ap.add_argument("--sorted-input", action="store_true", help="z-sort the cloud first (reported separately in DESIGN.md)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import treensearch_amd as T
    from treensearch_amd import datagen as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # TNSX_BENCH_FORCE_SLAB=1: exercise the slab / process-group path with a single rank (a 1-GPU box can check it)
    distributed = world > 1 or (os.environ.get("TNSX_BENCH_FORCE_SLAB") == "1" and "RANK" in os.environ)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU path)"
    torch.cuda.set_device(local_rank)
    if distributed:
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        assert args.gpus == 1, "launch N > 1 through torch.distributed.run"

    n = args.points
    # global cloud = `world` unit cubes side by side along x; rank k owns cube k (global ids k*n .. (k+1)*n-1)
    radius = D.radius_for_neighbors(n)            # ~60 neighbours at the per-slab density
    pts_h = D.uniform_cloud(n, args.seed, start=rank * n)
    pts_h[:, 0] += np.float32(rank)
    stream = torch.cuda.current_stream()
    arith = T.ARITH_STRICT if args.arith == "strict" else T.ARITH_CONTRACTED

    def make_engine():
        return T.TreeNSearch(arith=arith, stream=stream.cuda_stream, collect_stage_times=True, exact_layout=args.exact_layout)

    if distributed:
        from treensearch_amd.multi import SlabSearch
        gids = torch.arange(rank * n, (rank + 1) * n, dtype=torch.int64, device="cuda")
        slab = SlabSearch(float(rank), float(rank + 1), float(radius), make_engine)
        # the owned points live in the slab's own buffer, the ghosts of every step are appended behind them
        d_pts = slab.owned_buffer(n, "cuda", ghost_capacity=int(2.5 * n * float(radius)) + 4096)
        d_pts.copy_(torch.from_numpy(pts_h))
        ns = slab.engine

        def step():
            slab.step(d_pts, gids)
    else:
        ns = make_engine()
        ns.set_search_radius(radius)
        if args.sorted_input:
            tmp = T.TreeNSearch()
            tmp.set_search_radius(radius)
            tmp.add_point_set(pts_h)
            tmp.prepare_zsort()
            tmp.apply_zsort(0, pts_h, 3)
            del tmp
        d_pts = torch.from_numpy(pts_h).cuda()
        ns.add_point_set(d_pts)
        ns.set_active_search(0, 0, True)

        def step():
            ns.run()

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    acc = {}
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = ns.get_stats()
        for k in ("ms_total", "ms_bounds", "ms_keys", "ms_sort", "ms_gather", "ms_cells", "ms_count", "ms_scan", "ms_fill"):
            acc[k] = acc.get(k, 0.0) + st[k]
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    st = ns.get_stats()
    steps = max(args.steps, 1)
    ms_per_step = elapsed / steps * 1e3
    total_points = n * world
    value = total_points / (elapsed / steps) / 1e6

    # ---- roofline of the dominant kernel, this rank: the single-pass query (k_query_pool_fast; its two follow-up tiers
    #      run on empty worklists for this workload and are inside the same event bracket).  Algorithmic bytes per launch
    #      (DESIGN.md section 4): 16 B per candidate point read once (sorted float4) in, 4 B per emitted index + 4 B count
    #      word per query + 8 B per query (record offset by original index) out.  The exact two-pass layout
    #      (--exact-layout) additionally reads the 8 B scanned record offset per query in its fill pass.
    n_pts, Q, E = st["n_points"], st["n_queries"], st["n_neighbors"]
    pooled = st.get("n_pool_pairs", 0) > 0
    fill_bytes = 16 * n_pts + 4 * (E + Q) + 8 * Q + (0 if pooled else 8 * Q)
    fill_ms = acc["ms_fill"] / steps
    achieved = fill_bytes / (fill_ms * 1e-3) / 1e9 if fill_ms > 0 else 0.0
    traffic, traffic_detail = pmc_traffic(args.arith, pooled)
    run_bytes = st["bytes_build"] + st["bytes_query"]
    dev_ms = acc["ms_total"] / steps
    out = {
        "metric": "Mpoints/sec neighbor build+query", "value": round(value, 3), "unit": "Mpoints/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n} uniform-random points per GPU in a unit cube, single set, fixed radius "
                               f"r={float(radius):.6f} (~{E / max(Q, 1):.1f} neighbours avg), BASELINE.json configs[1]"
                               + ("" if world == 1 else f"; {world} slabs along x with one-radius ghost halos over RCCL"),
                   "points_per_gpu": n, "arith": args.arith, "input_order": "z-sorted" if args.sorted_input else "as generated (random)",
                   "neighbors_total_rank0": int(E), "grid": st["grid_dims"], "parallelism": f"slab{world}"},
        "roofline": {"bound": "hbm", "kernel": "k_query_pool_fast" if pooled else "k_query<fill>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_detail": traffic_detail,
                     "bytes_per_launch": int(fill_bytes), "avg_launch_ms": round(fill_ms, 4),
                     "whole_run": {"algorithmic_bytes": int(run_bytes), "bytes_per_point": round(run_bytes / max(n_pts, 1), 1),
                                   "device_ms": round(dev_ms, 4),
                                   "achieved_gbs": round(run_bytes / (dev_ms * 1e-3) / 1e9, 1) if dev_ms > 0 else 0.0,
                                   "frac": round(run_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dev_ms > 0 else 0.0}},
        "stage_ms": {k[3:]: round(v / steps, 4) for k, v in acc.items()},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, float(radius), args.seed)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
