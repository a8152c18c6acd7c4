"""bench.py's N > 1 control flow (configs[4]: cuts, redistribution and slab steps behind the C ABI, barrier + max-over-ranks timing, one JSON line from rank 0) as a
dry run on ONE GPU: two processes started the way the driver starts them (torch.distributed.run), sharing GPU 0, torch.distributed over gloo and the slab layer's
messages over the host-staged transport (TNSX_BENCH_SHARED_GPU=1: RCCL refuses two ranks on one device).  What is checked is that the path runs and that what it
reports is the same search: the neighbours per query of the decomposed cloud."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_two_ranks_on_a_shared_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TNSX_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--points", "2000000", "--no-cpu-baseline", "--no-pmc"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "strong" and d["config"]["name"] == "c5"
    assert d["config"]["points_total"] == 2000000 and len(d["config"]["cuts"]) == 1
    assert 0.4 < d["config"]["cuts"][0] < 0.6 and 900000 < d["config"]["points_rank0"] < 1100000
    assert 57.0 < d["config"]["neighbors_per_query"] < 61.0        # (the 2 M-point cloud of configs[4]'s scaled instance: 58.7 neighbours per point)
    assert d["steady_state"]["runs_repeated_after_a_failed_assumption"] <= 1 and d["value"] > 0
    # round 5: the line checks itself -- who exchanged what with whom, and what a speed-up is to be measured against
    tr = d["config"]["transport"]
    assert tr["ranks_by_allreduce_of_ones"] == 2, tr                      # every rank of the job was on the other end of the transport
    assert tr["kind"].startswith("application") and d["config"]["rccl_nranks"] is None     # (the dry run's transport is not RCCL, and the line says so)
    pr = d["config"]["per_rank"]
    assert len(pr["owned_points"]) == 2 and sum(pr["owned_points"]) == 2000000
    assert all(20000 < g < 120000 for g in pr["ghost_points_last_step"]), pr   # one halo of 1.11 r of a 1 M-point slab: ~57 k ghosts (capacity-padded rows included)
    assert all(ms > 0.0 for ms in pr["exchange_ms_per_step"]) and all(b > 0 for b in pr["bytes_sent_total"])
    assert d["one_gpu_same_workload"]["quoted"] is False                  # (no committed single-GPU run at 2 M points; at 200 M the line quotes profiles/bench_r4_c5_200m_1gpu.json)
