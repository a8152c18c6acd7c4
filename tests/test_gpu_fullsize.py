"""BASELINE.json configs[2] and configs[3] at FULL size (8 M + 2 M; 20 M dam break) through size-independent properties that hold
for any correct neighbour search, evaluated on the device (the lists never leave HBM): symmetry of the pair set (moments with
random 64-bit weights, wrapping arithmetic), self exclusion, distinct entries, index range, idempotence, invariance under the
z-sort; plus the configs[3] step -- perturb, prepare_zsort, apply_zsort(xyz), apply_zsort(radii), run -- against the oracle at a
size the oracle finishes in seconds."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import parity as P   # noqa: E402

pytestmark = pytest.mark.gpu


def _device_csr(ns, i, j):
    """(counts int64[n], owner int64[E], idx int64[E]) of pair (i, j) on the device; owner / idx are set-local indices"""
    import torch
    offs, idx = ns.neighbor_csr_torch(i, j)          # the device-side CSR of the public API (tnsx_pair_csr_device): nothing crosses the link
    counts = offs[1:] - offs[:-1]
    assert int(offs[-1].item()) == ns.pair_view(i, j).n_neighbors == idx.numel()
    owner = torch.repeat_interleave(torch.arange(counts.numel(), device="cuda", dtype=torch.int64), counts)
    return counts, owner, idx.to(torch.int64)


def _weights(n, seed):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randint(1, 1 << 62, (n,), generator=g, device="cuda", dtype=torch.int64)


def _moment(owner, idx, w_owner, w_idx):
    """sum over directed pairs (i, j) of w_owner[i] * w_idx[j]  (mod 2^64)"""
    return int((w_owner[owner] * w_idx[idx]).sum().item())


def _assert_lists_wellformed(owner, idx, n_j, same_set, what):
    import torch
    assert int(idx.min().item()) >= 0 and int(idx.max().item()) < n_j, f"{what}: index out of range"
    if same_set:
        assert not bool((idx == owner).any().item()), f"{what}: a point lists itself"
    key = torch.sort(owner * (1 << 31) + idx).values
    assert bool((key[1:] > key[:-1]).all().item()), f"{what}: duplicate entries inside a list"


def test_c3_full_size_properties():
    """8 M fluid + 2 M boundary.  0->0 is symmetric; 0->1 is checked against the reverse search 1->0 of a second engine (the roles of
    query and candidate set swapped); both are idempotent while the fluid stands still and the boundary build is served from the cache."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    f, b, r = D.two_set_cloud(8_000_000, 2_000_000)
    d_f, d_b = torch.from_numpy(f).cuda(), torch.from_numpy(b).cuda()
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(d_f)
    ns.add_point_set(d_b)
    ns.set_active_search(0, 0, True)
    ns.set_active_search(0, 1, True)
    ns.run()
    a0, b0, a1 = _weights(len(f), 1), _weights(len(f), 2), _weights(len(b), 3)
    c00, own00, idx00 = _device_csr(ns, 0, 0)
    assert 40 < float(c00.float().mean().item()) < 80
    _assert_lists_wellformed(own00, idx00, len(f), True, "0->0")
    m = _moment(own00, idx00, a0, b0)
    assert m == _moment(own00, idx00, b0, a0), "0->0: the pair set is not symmetric"
    c01, own01, idx01 = _device_csr(ns, 0, 1)
    _assert_lists_wellformed(own01, idx01, len(b), False, "0->1")
    m01 = _moment(own01, idx01, a0, a1)
    n01 = int(c01.sum().item())
    del own00, idx00, own01, idx01
    # the reverse search with another engine instance
    rev = T.TreeNSearch()
    rev.set_search_radius(r)
    rev.add_point_set(d_f)
    rev.add_point_set(d_b)
    rev.set_active_search(1, 0, True)
    rev.run()
    c10, own10, idx10 = _device_csr(rev, 1, 0)
    assert int(c10.sum().item()) == n01 and n01 > 1_000_000
    assert _moment(own10, idx10, a1, a0) == m01, "0->1 and 1->0 disagree on the pair set"
    del rev, own10, idx10
    # idempotence (second and third run: grid reused, boundary cached)
    for _ in range(2):
        ns.run()
    assert ns.get_stats()["n_cached_sets"] == 2 and ns.get_stats()["speculated"] == 1
    c00b, own, idx = _device_csr(ns, 0, 0)
    assert torch.equal(c00, c00b) and _moment(own, idx, a0, b0) == m
    c01b, own, idx = _device_csr(ns, 0, 1)
    assert torch.equal(c01, c01b) and _moment(own, idx, a0, a1) == m01


def test_c4_20m_properties_and_zsort_invariance():
    """20 M-point dam break, per-point radii, symmetric search: symmetric pair set, no self, distinct entries; after prepare_zsort +
    apply_zsort(xyz, radii, ids) the SAME pairs come out in terms of the points' identities."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 20_000_000
    p, rad, r0 = D.dam_break_cloud(n)
    d_p, d_r = torch.from_numpy(p).cuda(), torch.from_numpy(rad).cuda()
    ids = torch.arange(n, dtype=torch.int64, device="cuda")
    ns = T.TreeNSearch()
    ns.add_point_set(d_p, d_r)
    ns.set_active_search(0, 0, True)
    ns.set_symmetric_search(True)
    ns.run()
    a, b = _weights(n, 11), _weights(n, 12)
    cnt, own, idx = _device_csr(ns, 0, 0)
    _assert_lists_wellformed(own, idx, n, True, "c4")
    m_ab = _moment(own, idx, a, b)
    assert m_ab == _moment(own, idx, b, a), "symmetric search, yet the pair set is not symmetric"
    total = int(cnt.sum().item())
    deg = int((a * cnt).sum().item())                           # identity-weighted degree sum
    del own, idx
    ns.prepare_zsort()
    ns.apply_zsort(0, d_p, 3)
    ns.apply_zsort(0, d_r, 1)
    ns.apply_zsort(0, ids, 1)
    assert not torch.equal(ids, torch.arange(n, dtype=torch.int64, device="cuda"))
    ns.run()
    cnt2, own2, idx2 = _device_csr(ns, 0, 0)
    assert int(cnt2.sum().item()) == total
    assert int((a[ids] * cnt2).sum().item()) == deg, "neighbour counts changed under the z-sort"
    assert _moment(own2, idx2, a[ids], b[ids]) == m_ab, "the pair set changed under the z-sort"


def test_c4_step_loop_matches_oracle(oracle):
    """configs[3], five steps at 100 k points: perturb (<= 0.1 r0), prepare_zsort, apply_zsort(xyz), apply_zsort(radii), run -- every step
    against the oracle on the arrays as they are after the permutation; symmetric and asymmetric."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 100_000
    for symmetric in (True, False):
        p, rad, r0 = D.dam_break_cloud(n, 21)
        d_p, d_r = torch.from_numpy(p).cuda(), torch.from_numpy(rad).cuda()
        ns = T.TreeNSearch()
        ns.add_point_set(d_p, d_r)
        ns.set_active_search(0, 0, True)
        ns.set_symmetric_search(symmetric)
        g = torch.Generator(device="cuda").manual_seed(7)
        for step in range(5):
            d_p.add_((torch.rand(d_p.shape, generator=g, device="cuda") - 0.5) * (2.0 * 0.1 * float(r0) / 3.0 ** 0.5))
            ns.prepare_zsort()
            ns.apply_zsort(0, d_p, 3)
            ns.apply_zsort(0, d_r, 1)
            ns.run()
            hp, hr = d_p.cpu().numpy(), d_r.cpu().numpy()
            ref = oracle.pair_search(hp, hp, ra=hr, rb=hr, symmetric=symmetric, same_set=True)
            P.assert_same_csr(ns.neighbor_csr(0, 0), ref, f"c4 loop step {step} symmetric={symmetric}")
            # the order handed out is a permutation and the points really are z-sorted on the reference grid afterwards
            order = ns.get_zsort_order(0)
            assert np.array_equal(np.sort(order), np.arange(n))


# ---------------------------------------------------------------------------------------------------------------------
# configs[3] at its full 50 M points and configs[4] at its full 200 M points (one rank, through the slab layer's C entry points).
# The same size-independent properties, evaluated on the device in chunks of query points so that no temporary is larger than a
# few GB: index range, no self, distinct entries, symmetry of the pair set by random-weight moments, z-sort invariance (c4).
# ---------------------------------------------------------------------------------------------------------------------
def _chunked_properties(ns, i, j, n_j, same_set, w_a, w_b, chunk=4_000_000, owner_ids=None):
    """-> (total, sum w_a[i] w_b[j], sum w_b[i] w_a[j], sum w_a[i] count_i) over all directed pairs, all mod 2^64; asserts
    well-formedness of every list on the way.  owner_ids: identity of query point p (default p); list entries are identities."""
    import torch
    v = ns.pair_view(i, j)
    offs, recs = ns.neighbor_records_torch(i, j)     # zero-copy views of the engine's own offsets / records (at 200 M points the records are 48 GB)
    total, m_ab, m_ba, deg = 0, 0, 0, 0
    M = (1 << 64) - 1
    for lo in range(0, v.n_points, chunk):
        hi = min(lo + chunk, v.n_points)
        o = offs[lo:hi]
        counts = recs[o].to(torch.int64)
        e = int(counts.sum().item())
        start = torch.cumsum(counts, 0) - counts
        src = torch.repeat_interleave(o + 1 - start, counts) + torch.arange(e, device="cuda", dtype=torch.int64)
        idx = recs[src].to(torch.int64)
        del src
        local = torch.repeat_interleave(torch.arange(hi - lo, device="cuda", dtype=torch.int64), counts)
        owner = (local + lo) if owner_ids is None else owner_ids[lo:hi][local]
        assert int(idx.min().item()) >= 0 and int(idx.max().item()) < n_j, "index out of range"
        if same_set:
            assert not bool((idx == owner).any().item()), "a point lists itself"
        key = torch.sort(local * (1 << 31) + idx).values
        assert bool((key[1:] > key[:-1]).all().item()), "duplicate entries inside a list"
        del key, local
        total += e
        m_ab = (m_ab + int((w_a[owner] * w_b[idx]).sum().item())) & M
        m_ba = (m_ba + int((w_b[owner] * w_a[idx]).sum().item())) & M
        own_id = torch.arange(lo, hi, device="cuda", dtype=torch.int64) if owner_ids is None else owner_ids[lo:hi]
        deg = (deg + int((w_a[own_id] * counts).sum().item())) & M
        del idx, owner, counts
    assert total == v.n_neighbors
    return total, m_ab, m_ba, deg


def _sampled_lists(ns, i, j, sample):
    """Sorted neighbour lists of the query points `sample` (int64 numpy, set-local indices of set i) of pair (i, j), fetched from the device copy of the
    pair: (offsets int64[k + 1], indices int32[E]) like oracle.pair_search returns them."""
    import torch
    offs, recs = ns.neighbor_records_torch(i, j)
    o = offs[torch.from_numpy(sample).cuda()]
    counts = recs[o].to(torch.int64)
    e = int(counts.sum().item())
    start = torch.cumsum(counts, 0) - counts
    src = torch.repeat_interleave(o + 1 - start, counts) + torch.arange(e, device="cuda", dtype=torch.int64)
    idx = recs[src].to(torch.int64)
    local = torch.repeat_interleave(torch.arange(len(sample), device="cuda", dtype=torch.int64), counts)
    idx = (torch.sort(local * (1 << 31) + idx).values & ((1 << 31) - 1)).to(torch.int32).cpu().numpy()
    out_offs = np.zeros(len(sample) + 1, np.int64)
    out_offs[1:] = np.cumsum(counts.cpu().numpy())
    return out_offs, idx


def _assert_sample_equals_all_points_search(oracle, ns, i, j, sample, pts, what, radii=None, radius=None, symmetric=True):
    """Round-4 verdict item 4: EXACT parity at full size on a seeded sample.  The sample's points are searched in ALL points by the CPU restatement
    (BruteforceNSearch.cpp:78-100 semantics: every candidate that passes the predicate, nothing else decides) as a pair of two different sets, the point
    itself is taken out by its global index (a search of a set in itself never lists the point, TreeNSearch.cpp:2468), and the result must equal the
    engine's lists of those points entry by entry."""
    xa = np.ascontiguousarray(pts[sample])
    if radii is not None:
        ro, ri = oracle.pair_search(xa, pts, ra=np.ascontiguousarray(radii[sample]), rb=radii, symmetric=symmetric, same_set=False)
    else:
        ro, ri = oracle.pair_search(xa, pts, radius=radius, same_set=False)
    owner = np.repeat(np.arange(len(sample)), np.diff(ro))
    keep = ri != sample[owner].astype(np.int32)
    assert int((~keep).sum()) == len(sample), f"{what}: every sampled point finds itself exactly once in the all-points search"
    ri = ri[keep]
    want_offs = np.zeros(len(sample) + 1, np.int64)
    want_offs[1:] = np.cumsum(np.bincount(owner[keep], minlength=len(sample)))
    got_offs, got_idx = _sampled_lists(ns, i, j, sample)
    assert np.array_equal(got_offs, want_offs), f"{what}: neighbour counts of the sampled points differ"
    assert np.array_equal(got_idx, ri), f"{what}: neighbour lists of the sampled points differ"
    return int(want_offs[-1])


def test_c4_full_size_50m_properties_and_zsort_invariance(oracle):
    """BASELINE.json configs[3] at its full size: 50 M-point dam break, per-point radii, symmetric search.  Size-independent properties of ALL lists, and
    (round 5) the lists of 20 000 randomly drawn points EXACTLY against an all-points search of the CPU restatement."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 50_000_000
    p, rad, r0 = D.dam_break_cloud(n)
    d_p, d_r = torch.from_numpy(p).cuda(), torch.from_numpy(rad).cuda()
    ids = torch.arange(n, dtype=torch.int64, device="cuda")
    ns = T.TreeNSearch()
    ns.add_point_set(d_p, d_r)
    ns.set_active_search(0, 0, True)
    ns.set_symmetric_search(True)
    ns.run()
    a, b = _weights(n, 11), _weights(n, 12)
    total, m_ab, m_ba, deg = _chunked_properties(ns, 0, 0, n, True, a, b)
    assert 55 * n < total < 70 * n
    assert m_ab == m_ba, "symmetric search, yet the pair set is not symmetric"
    sample = np.sort(np.random.default_rng(505).choice(n, 20000, replace=False)).astype(np.int64)
    e = _assert_sample_equals_all_points_search(oracle, ns, 0, 0, sample, p, "c4 at 50 M", radii=rad, symmetric=True)
    assert 50 * len(sample) < e < 75 * len(sample)
    del p, rad
    ns.prepare_zsort()
    ns.apply_zsort(0, d_p, 3)
    ns.apply_zsort(0, d_r, 1)
    ns.apply_zsort(0, ids, 1)
    ns.run()
    # identities: query point p is ids[p], entry j is ids[j]
    total2, m_ab2, m_ba2, deg2 = _chunked_properties(ns, 0, 0, n, True, a[ids], b[ids])
    # (weights permuted to positions: w'[p] = w[ids[p]] -- the moments are then sums over identities)
    assert total2 == total and m_ab2 == m_ab and m_ba2 == m_ba, "the pair set changed under the z-sort"
    assert deg2 == deg, "neighbour counts changed under the z-sort"


def test_c5_full_size_200m_one_rank_through_the_slab_layer(oracle):
    """BASELINE.json configs[4] at its full 200 M points on ONE GPU, through tnsx_slab_step (one rank: no exchange, but the whole slab
    path -- [owned | ghosts] buffers, candidates-only tail, global ids from the engine): the lists hold global ids; symmetric pair set,
    no self, distinct entries, ~59 neighbours per point, idempotent second step; (round 5) the lists of 20 000 randomly drawn points EXACTLY
    against an all-points search of the CPU restatement (the lists hold GLOBAL ids: here the identity)."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    from treensearch_amd.multi import SlabSearchC
    n = 200_000_000
    r = D.radius_for_neighbors(n)
    pts = D.uniform_cloud(n, 12345)
    d_p = torch.from_numpy(pts).cuda()
    gids = torch.arange(n, dtype=torch.int64, device="cuda")
    ns = T.TreeNSearch()
    slab = SlabSearchC(float("-inf"), float("inf"), float(r), ns, None, 0, 1)
    slab.step(d_p, gids)
    a, b = _weights(n, 21), _weights(n, 22)
    sid = slab.set_id(0)
    total, m_ab, m_ba, deg = _chunked_properties(ns, sid, sid, n, True, a, b, chunk=5_000_000)
    assert 58.5 * n < total < 60.5 * n
    assert m_ab == m_ba, "the pair set is not symmetric"
    sample = np.sort(np.random.default_rng(606).choice(n, 20000, replace=False)).astype(np.int64)
    e = _assert_sample_equals_all_points_search(oracle, ns, sid, sid, sample, pts, "c5 at 200 M", radius=float(r))
    assert 55 * len(sample) < e < 64 * len(sample)
    del pts
    slab.step(d_p, gids)
    st = ns.get_stats()
    assert st["n_neighbors"] == total and st["speculated"] == 1


def test_sparse_grid_10m_filament(oracle):
    """SURVEY.md section 8 / round-3 verdict item 8 at its stated size: a 10 M-point filament that winds through the whole box -- a grid of ~3 x 10^9 cells of
    one search radius, 0.2 % of them occupied.  With sparse_grid = 1 the cells keep their edge (tnsx_stats.grid_sparse: lists of occupied cells + block index
    instead of a dense table) and the lists equal the CPU restatement's, by count per point and by the order-independent digest; the coarse dense grid of
    rounds 1-3 (sparse_grid = -1) gives the same digest and is timed beside it.  (The grid is three times the dense bound: the default would coarsen the
    cells by 1.6 -- the faster choice for a cloud this thin, DESIGN.md section 9 -- and switches to the sparse grid from eight times on.)"""
    import time
    import torch
    import treensearch_amd as T
    n = 10_000_000
    rng = np.random.default_rng(3)
    t = np.sort(rng.random(n))
    turns = 420.0
    ang = 2.0 * np.pi * turns * t
    r = np.float32(0.00075)
    pts = np.stack([0.5 + 0.45 * np.cos(ang), 0.5 + 0.45 * np.sin(ang), 0.02 + 0.96 * t], axis=1)
    pts += (rng.random((n, 3)) - 0.5) * (0.6 * float(r))
    pts = np.ascontiguousarray(pts.astype(np.float32))
    d = torch.from_numpy(pts).cuda()
    times = {}
    res = {}
    for name, kw in (("sparse", {"sparse_grid": 1}), ("coarse", {"sparse_grid": -1})):
        ns = T.TreeNSearch(**kw)
        ns.set_search_radius(r)
        ns.add_point_set(d)
        ns.set_active_search(0, 0, True)
        ns.run(); ns.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ns.run()
        times[name] = (time.perf_counter() - t0) / 3 * 1e3
        st = ns.get_stats()
        if name == "sparse":
            assert st["grid_sparse"] == 1 and abs(st["grid_cell_size"] / float(r) - 1.0) < 1e-3, (st["grid_sparse"], st["grid_cell_size"])
            assert st["n_occupied_cells"] * 100 < st["n_grid_cells"], "less than 1 % of the grid is occupied"
        else:
            assert st["grid_sparse"] == 0 and st["grid_cell_size"] > 1.2 * float(r)
        offs, idx = ns.neighbor_csr(0, 0, sort_each=False)
        res[name] = (offs, oracle.digest(offs, idx, already_sorted=False))
        del ns
    ro, ri = oracle.pair_search(pts, pts, radius=r, same_set=True)
    want = oracle.digest(ro, ri, already_sorted=True)
    for name in res:
        assert np.array_equal(res[name][0], ro), f"{name}: neighbour counts differ from the CPU restatement"
        assert res[name][1] == want, f"{name}: lists differ from the CPU restatement"
    print(f"10 M-point filament, {int(ro[-1])} neighbours: sparse grid {times['sparse']:.2f} ms per run, coarse dense grid {times['coarse']:.2f} ms")
