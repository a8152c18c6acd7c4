"""Multi-process tests (gloo, CPU) of the slab layer in treensearch_amd/multi.py: decomposition (global AABB, x histogram,
balanced cuts, redistribution), the ghost-halo exchange and the [owned | ghosts] search with global ids.

The product has no CPU search path, so the per-rank search backend is injected: a stand-in with the TreeNSearch API
whose run() calls the CPU oracle (tests/slab_helpers.py).  What is under test is the distributed logic: where the cuts are,
which points travel, the 64-bit global-id transport, candidates-only ghosts and global ids in the lists -- the union of the
ranks' results must equal the single-process result on the whole cloud.  tests/test_gpu_slabs.py runs the same SlabSearch
code on the HIP engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from slab_helpers import OracleEngine, reference_lists   # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


# ----------------------------------------------------------------------------------------------------------------------
# fixed slabs (unit cubes side by side), single set, fixed radius: exact one-round / two-round protocol of the exchange
# ----------------------------------------------------------------------------------------------------------------------
def _worker_fixed(rank, world, port, n_per_rank, radius, tmpdir):
    _init(rank, world, port)
    try:
        from treensearch_amd import datagen as D
        from treensearch_amd.multi import SlabSearch
        pts = D.uniform_cloud(n_per_rank, 4711, start=rank * n_per_rank)
        pts[:, 0] += np.float32(rank)                      # slab k = unit cube [k, k+1) x [0,1)^2
        gids = torch.arange(rank * n_per_rank, (rank + 1) * n_per_rank, dtype=torch.int64)
        slab = SlabSearch(float(rank), float(rank + 1), float(radius), OracleEngine)
        t_pts = torch.from_numpy(pts)
        rounds = []
        for step in range(4):                              # step 0: two rounds (no capacity agreed yet); step 1: one round;
            if step == 2:                                  # step 2: twice the halo => overflow => two rounds again; step 3: back to the
                slab.ex.halo *= 2.0                        # original halo, one round (the capacity only grows)
            if step == 3:
                slab.ex.halo /= 2.0
            slab.step(t_pts, gids)
            rounds.append(slab.ex.rounds_last)
        assert rounds == [2, 1, 2, 1], rounds
        offs, nbr = slab.global_neighbors()
        np.save(os.path.join(tmpdir, f"offs_{rank}.npy"), offs)
        np.save(os.path.join(tmpdir, f"nbr_{rank}.npy"), nbr)
        np.save(os.path.join(tmpdir, f"nghost_{rank}.npy"), np.array([len(slab.ghost_gids), slab.ex.bytes_sent]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_halo_exchange_matches_single_process(world, tmp_path, oracle):
    from treensearch_amd import datagen as D
    n_per_rank, radius = 4000, np.float32(0.09)
    mp.spawn(_worker_fixed, args=(world, _free_port(), n_per_rank, radius, str(tmp_path)), nprocs=world, join=True)
    # single-process truth on the union of all slabs
    allp = []
    for k in range(world):
        p = D.uniform_cloud(n_per_rank, 4711, start=k * n_per_rank)
        p[:, 0] += np.float32(k)
        allp.append(p)
    allp = np.concatenate(allp)
    offs, idx = oracle.pair_search(allp, allp, radius=radius, same_set=True)
    for k in range(world):
        o = np.load(tmp_path / f"offs_{k}.npy")
        nb = np.load(tmp_path / f"nbr_{k}.npy")
        lo = k * n_per_rank
        ref_o = offs[lo:lo + n_per_rank + 1] - offs[lo]
        ref_i = idx[offs[lo]:offs[lo + n_per_rank]].astype(np.int64)
        assert np.array_equal(o, ref_o), f"rank {k}: neighbour counts differ"
        assert np.array_equal(nb, ref_i), f"rank {k}: global neighbour ids differ"
        n_ghost, _ = np.load(tmp_path / f"nghost_{k}.npy")
        assert n_ghost > 0
        # only a thin halo travels: ~radius/1.0 of each neighbouring slab
        n_faces = (1 if k > 0 else 0) + (1 if k < world - 1 else 0)
        assert n_ghost < n_faces * n_per_rank * float(radius) * 1.5


# ----------------------------------------------------------------------------------------------------------------------
# the whole pipeline: arbitrary initial distribution -> global AABB -> histogram -> balanced cuts -> redistribution ->
# slab search; clustered cloud (unequal cuts), per-point radii, two sets with asymmetric searches
# ----------------------------------------------------------------------------------------------------------------------
def _cloud(kind: str, n: int):
    """(sets, fixed radius or None, max radius): sets = [(points, radii or None)], the GLOBAL cloud (same on every rank)"""
    from treensearch_amd import datagen as D
    if kind == "uniform":
        return [(D.uniform_cloud(n, 99), None)], D.radius_for_neighbors(n, 30.0), None
    if kind == "clustered_var":
        pts, radii, r0 = D.dam_break_cloud(n, 7, neighbors=25.0)
        return [(pts, radii)], None, float(2.0 * r0)
    if kind == "two_sets":
        f, b, r = D.two_set_cloud(int(0.8 * n), n - int(0.8 * n), 5)
        return [(f, None), (b, None)], np.float32(1.2 * float(r) * (60.0 / 30.0) ** (-1 / 3)), None
    raise ValueError(kind)


def _worker_pipeline(rank, world, port, kind, n, tmpdir):
    _init(rank, world, port)
    try:
        from treensearch_amd.multi import SlabDecomposition, SlabSearch
        sets, radius, max_radius = _cloud(kind, n)
        halo_r = float(radius) if radius is not None else max_radius
        # initial distribution: a contiguous index range per rank (nothing to do with space)
        mine = []
        for (p, r) in sets:
            lo, hi = (len(p) * rank) // world, (len(p) * (rank + 1)) // world
            mine.append((torch.from_numpy(p[lo:hi]), torch.arange(lo, hi, dtype=torch.int64), None if r is None else torch.from_numpy(r[lo:hi])))
        dec = SlabDecomposition()
        bounds = dec.global_bounds([m[0] for m in mine])
        cuts = dec.balanced_cuts([m[0] for m in mine], plane_width=halo_r * 1.001, bounds=bounds)
        owned = [dec.redistribute(pp, gg, rr, cuts) for (pp, gg, rr) in mine]
        slab = SlabSearch(float(cuts[rank]), float(cuts[rank + 1]), None if radius is None else float(radius), OracleEngine, max_radius=max_radius)
        pairs = [(0, 0)] if len(sets) == 1 else [(0, 0), (0, 1)]
        for (i, j) in pairs:
            slab.set_active_search(i, j, True)
        for _ in range(2):
            slab.step(*[(pp, gg) + ((rr,) if rr is not None else ()) for (pp, gg, rr) in owned])
        for (i, j) in pairs:
            offs, nbr = slab.global_neighbors(i, j)
            np.save(os.path.join(tmpdir, f"offs_{i}{j}_{rank}.npy"), offs)
            np.save(os.path.join(tmpdir, f"nbr_{i}{j}_{rank}.npy"), nbr)
        for k, (pp, gg, rr) in enumerate(owned):
            np.save(os.path.join(tmpdir, f"gids_{k}_{rank}.npy"), gg.numpy())
        np.save(os.path.join(tmpdir, f"cuts_{rank}.npy"), cuts)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,n", [(8, "uniform", 24000), (4, "clustered_var", 20000), (3, "two_sets", 15000)])
def test_decomposition_pipeline_matches_single_process(world, kind, n, tmp_path, oracle):
    mp.spawn(_worker_pipeline, args=(world, _free_port(), kind, n, str(tmp_path)), nprocs=world, join=True)
    sets, radius, max_radius = _cloud(kind, n)
    pairs = [(0, 0)] if len(sets) == 1 else [(0, 0), (0, 1)]
    cuts = np.load(tmp_path / "cuts_0.npy")
    for k in range(1, world):
        assert np.array_equal(cuts, np.load(tmp_path / f"cuts_{k}.npy")), "the ranks disagree on the cuts"
    assert np.all(np.diff(cuts[1:-1]) > 0)
    # balance: every slab of set 0 holds about 1/world of the points (plane granularity)
    owned0 = [np.load(tmp_path / f"gids_0_{k}.npy") for k in range(world)]
    sizes = np.array([len(g) for g in owned0])
    assert sizes.sum() == len(sets[0][0]) and len(np.unique(np.concatenate(owned0))) == len(sets[0][0]), "points lost or duplicated"
    x = sets[0][0][:, 0]
    for k in range(world):
        assert np.all((x[owned0[k]] >= cuts[k]) & (x[owned0[k]] < cuts[k + 1])), f"rank {k} owns points outside its slab"
    if kind == "uniform":
        assert sizes.max() < 1.35 * sizes.mean(), sizes
    else:
        assert len(set(np.round(np.diff(cuts[1:-1]), 6))) > 1 or world <= 3, "clustered cloud, yet equidistant cuts"
    # lists
    for (i, j) in pairs:
        ref_o, ref_i = reference_lists(oracle, sets, i, j, radius)
        cnt = np.diff(ref_o)
        for k in range(world):
            gids = np.load(tmp_path / f"gids_{i}_{k}.npy")
            o = np.load(tmp_path / f"offs_{i}{j}_{k}.npy")
            nb = np.load(tmp_path / f"nbr_{i}{j}_{k}.npy")
            assert np.array_equal(np.diff(o), cnt[gids]), f"pair {i}->{j}, rank {k}: neighbour counts differ"
            want = np.concatenate([ref_i[ref_o[g]:ref_o[g + 1]] for g in gids]) if len(gids) else np.zeros(0, np.int64)
            assert np.array_equal(nb, want.astype(np.int64)), f"pair {i}->{j}, rank {k}: global neighbour ids differ"


# ----------------------------------------------------------------------------------------------------------------------
# single-process pieces
# ----------------------------------------------------------------------------------------------------------------------
def test_halo_masks():
    from treensearch_amd.multi import slab_halo_masks
    x = torch.tensor([0.0, 0.04, 0.5, 0.96, 0.999])
    left, right = slab_halo_masks(x, 0.0, 1.0, 0.05, True, True)
    assert left.tolist() == [True, True, False, False, False]
    assert right.tolist() == [False, False, False, True, True]
    left, right = slab_halo_masks(x, 0.0, 1.0, 0.05, False, True)
    assert not left.any() and right.sum() == 2


def test_single_rank_has_no_ghosts():
    """No process group / one rank: nothing to exchange, and the empty ghost arrays must still be well-formed."""
    from treensearch_amd.multi import SlabExchange, SlabSearch
    pts = torch.rand(100, 3)
    gids = torch.arange(100, dtype=torch.int64)
    for radii in (None, torch.full((100,), 0.1)):
        gp, gg, gr = SlabExchange(0.0, 1.0, 0.1).exchange(pts, gids, radii)
        assert gp.shape == (0, 3) and gg.shape == (0,) and gg.dtype == torch.int64
        assert (gr is None) == (radii is None)
    slab = SlabSearch(0.0, 1.0, 0.2, OracleEngine)
    view = slab.owned_buffer(100, "cpu")
    view.copy_(pts)
    slab.step(view, gids)
    slab.step(view, gids)
    offs, nbr = slab.global_neighbors()
    from oracle import oracle as O
    ro, ri = O.Oracle().pair_search(pts.numpy(), pts.numpy(), radius=np.float32(0.2), same_set=True)
    assert np.array_equal(offs, ro) and np.array_equal(nbr, ri.astype(np.int64))


def test_balanced_cuts_single_process():
    """world 1 has no cuts; the histogram helper agrees with numpy."""
    from treensearch_amd.multi import SlabDecomposition
    from treensearch_amd import datagen as D
    pts = torch.from_numpy(D.uniform_cloud(5000, 3))
    dec = SlabDecomposition()
    lo, hi = dec.global_bounds([pts])
    assert np.allclose(lo, pts.numpy().min(0)) and np.allclose(hi, pts.numpy().max(0))
    h = dec.x_histogram([pts], float(lo[0]), 0.05, 20).numpy()
    want = np.bincount(np.clip(((pts.numpy()[:, 0] - lo[0]) * np.float32(1.0 / np.float32(0.05))).astype(np.int64), 0, 19), minlength=20)
    assert np.array_equal(h, want)
    cuts = dec.balanced_cuts([pts], 0.05)
    assert len(cuts) == 2 and cuts[0] == -np.inf and cuts[1] == np.inf
    own = SlabDecomposition.owner_of(torch.tensor([0.1, 0.5, 0.9]), np.array([-np.inf, 0.3, 0.6, np.inf], np.float32))
    assert own.tolist() == [0, 1, 2]
