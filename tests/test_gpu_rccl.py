"""The slab layer over RCCL with MORE THAN ONE rank -- `rccl_exchange` / `rccl_allreduce` of treensearch_amd/csrc/tnsx_slab.cpp, the grouped
ncclSend / ncclRecv of every step -- and the multi-device mode of the C ABI on DIFFERENT devices (TNSX_DEVICES=0,1).

RCCL refuses two ranks on one device, so these tests need a box that shows at least two devices: an 8-GPU node, or ONE MI355X in CPX compute-partition
mode (8 XCDs -> 8 logical devices; tools/jobs/r6_cpx.sh tries to switch it on).  They skip everywhere else.  Not a scaling measurement on a partitioned
GPU (same silicon): an execution of the code path, with the union of the ranks' lists checked against the reference's digest.
"""
import os
import socket

import numpy as np
import pytest

import cases as CS
from conftest import load_golden
from slab_helpers import union_csr

pytestmark = pytest.mark.gpu


def _n_devices() -> int:
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_n_devices() < 2, reason="needs >= 2 devices (an 8-GPU node or one MI355X in CPX mode): RCCL refuses two ranks on one device")


def _rccl_worker(rank, world, port, case_name, out_dir):
    """one process per rank, rank k on device k; torch.distributed (gloo) only carries the 128-byte unique id, every message of the slab layer goes
    through libtnsx.so's own ncclSend / ncclRecv / ncclAllReduce"""
    import torch
    import torch.distributed as dist
    import treensearch_amd as T
    from treensearch_amd.multi import SlabSearchC, SlabTransportC, balanced_cuts_c, redistribute_c, transport_check_c
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank)
    case = CS.by_name(case_name)
    p_h = case.points[0]
    n = len(p_h)
    lo_i, hi_i = (n * rank) // world, (n * (rank + 1)) // world
    eng = T.TreeNSearch()
    tr = SlabTransportC.rccl(rank, world, device=rank)
    seen = transport_check_c(eng, tr, rank, world)
    share = torch.from_numpy(np.ascontiguousarray(p_h[lo_i:hi_i])).cuda()
    cuts = balanced_cuts_c(eng, tr, rank, world, [share], float(case.radius) * 1.002)
    pts, gids = redistribute_c(eng, tr, rank, world, cuts, share, torch.arange(lo_i, hi_i, dtype=torch.int64, device="cuda"))
    slab = SlabSearchC(float(cuts[rank]), float(cuts[rank + 1]), float(case.radius), eng, tr, rank, world)
    slab.set_watchdog(60.0)
    inf0 = slab.info()
    log = []
    for _ in range(3):
        slab.step(pts, gids)
        inf = slab.info()
        log.append((int(inf.speculative_last), int(inf.redone_last), int(inf.rounds_last)))
    offs, idx = eng.neighbor_csr(slab.set_id(0), slab.set_id(0), sort_each=False)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), gids=gids.cpu().numpy(), offs=offs, idx=idx.astype(np.int64), cuts=cuts, log=np.array(log),
             seen=np.array([seen, int(inf0.transport_kind), int(inf0.transport_ranks), int(inf.n_ghost)]))
    del slab
    dist.barrier()
    dist.destroy_process_group()


@needs_two
@pytest.mark.parametrize("world", [2, 4, 8])
def test_slab_union_over_rccl(world, oracle, tmp_path):
    """cuts (two all-reduces), the redistribution (one all-to-all of grouped sends) and three steps (exact, then speculative single rounds) of the 2 M-point
    scaled instance of configs[4] over RCCL; union of the ranks' lists == the reference's digest"""
    import torch.multiprocessing as mp
    if _n_devices() < world:
        pytest.skip(f"{_n_devices()} devices")
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    name = "uniform_fixed_2000000"
    mp.start_processes(_rccl_worker, args=(world, port, name, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    case = CS.by_name(name)
    per_rank, cuts0 = [], None
    for k in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{k}.npz"))
        per_rank.append((d["gids"], d["offs"], d["idx"]))
        cuts0 = d["cuts"] if cuts0 is None else cuts0
        assert np.array_equal(cuts0, d["cuts"]), "every rank must arrive at the same cuts"
        seen, kind, ranks, ghosts = (int(x) for x in d["seen"])
        assert seen == world and kind == 1 and ranks == world, f"rank {k}: all-reduce of ones {seen}, transport kind {kind} (1 = RCCL), ncclCommCount {ranks}"
        assert ghosts > 0
        log = [tuple(x) for x in d["log"]]
        assert log[0] == (0, 0, 2) and log[1] == (1, 0, 1) and log[2] == (1, 0, 1), f"rank {k}: exact step first, then single speculative rounds; got {log}"
    assert sum(len(g) for g, _, _ in per_rank) == len(case.points[0])
    g_offs, g_idx = union_csr(len(case.points[0]), per_rank)
    fx = load_golden(case.name)["pairs"]["0->0"]["strict"]
    d_union = oracle.digest(g_offs, g_idx.astype(np.int32))
    assert int(g_offs[-1]) == fx["total"]
    assert (f"{d_union[0]:016x}", f"{d_union[1]:016x}") == (fx["digest_sum"], fx["digest_xor"]), "union of the ranks' lists differs from the reference's digest"


@needs_two
@pytest.mark.parametrize("name", ["uniform_fixed_100000", "two_set_asym_80000_20000", "dam_break_sym_100000"])
def test_multi_device_context_on_different_devices(name, oracle):
    """tnsx_options.n_devices with engines on DIFFERENT devices (what TNSX_DEVICES=0,1 gives the C++ drop-in): one link and one engine per slab"""
    import parity as P
    import treensearch_amd as T
    case = CS.by_name(name)
    devs = list(range(min(_n_devices(), 4)))
    ns = T.TreeNSearch(devices=devs)
    variable = case.radii is not None
    if not variable:
        ns.set_search_radius(case.radius)
    for s, p in enumerate(case.points):
        ns.add_point_set(p, case.radii[s] if variable else None)
    for (i, j) in case.active:
        ns.set_active_search(i, j, True)
    ns.set_symmetric_search(case.symmetric)
    for step in range(2):
        ns.run()
        res = {pr: ns.neighbor_csr(*pr) for pr in case.active}
        P.assert_matches_golden(res, load_golden(case.name), 0, oracle, f"{name} on devices {devs} (run {step})")
    assert ns.get_stats()["n_devices_used"] == len(devs)
