"""Temporal reuse (TreeNSearch.cpp:474-482 / :77-79): a run lays the previous run's grid over the points without
computing their bounds first, and unchanged sets keep their search structures.  Both are speculation that the device verifies
during the run; these tests drive the engine through the cases where the speculation holds and where it must be repaired,
and compare every step with the oracle (lists) and with the oracle's restatement of the reference's world-box rule."""
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

F32_MAX = np.finfo(np.float32).max


def _oracle_box(oracle, box, pts_list, cell):
    tight = np.array([F32_MAX] * 3 + [-F32_MAX] * 3, np.float32)
    for p in pts_list:
        if len(p):
            tight[:3] = np.minimum(tight[:3], p.min(axis=0))
            tight[3:] = np.maximum(tight[3:], p.max(axis=0))
    if np.all(tight[:3] <= tight[3:]):
        # run() sees the tight box united with the origin (TreeNSearch.cpp:564-569 + :587-590; pinned by the `world` fixtures)
        tight[:3] = np.minimum(tight[:3], np.float32(0.0))
        tight[3:] = np.maximum(tight[3:], np.float32(0.0))
    oracle.world_box_update(box, tight, cell)
    return box


@pytest.mark.parametrize("on_device", [True, False])
def test_moving_cloud_grid_reuse_and_world_box(on_device, oracle):
    """A cloud that drifts and expands step by step.  Steps inside the margin of the grid reuse it (speculated, no redo); the
    step that leaves the box is noticed on the device and repeated; lists are exact at every step and the world box follows the
    reference's rule (kept while it contains the points, re-snapped otherwise)."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 60000
    base = D.uniform_cloud(n, 31) * np.float32(0.5) + np.float32(0.25)
    r = D.radius_for_neighbors(n, 35.0, 0.125)
    pts = base.copy()
    d_pts = torch.from_numpy(pts).cuda() if on_device else None
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(d_pts if on_device else pts)
    ns.set_active_search(0, 0, True)
    box = np.array([F32_MAX] * 3 + [-F32_MAX] * 3, np.float32)
    rng = np.random.default_rng(3)
    seen = {"reused": 0, "redone": 0}
    #        small jitter (stays in the margin) ... one point shoots out ... jitter ... the whole cloud scales up beyond the world box
    moves = ["jitter", "jitter", "escape", "jitter", "jitter", "blow_up", "jitter"]
    for step, mv in enumerate(["first"] + moves):
        if mv == "jitter":
            pts += (rng.random(pts.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(0.2) * r
        elif mv == "escape":
            pts[12345] = pts.max(axis=0) + np.float32(3.5) * r    # beyond the two-radius margin of the grid
        elif mv == "blow_up":
            pts[:] = (pts - np.float32(0.5)) * np.float32(1.6) + np.float32(0.5)
        if on_device:
            d_pts.copy_(torch.from_numpy(pts))
        ns.run()
        st = ns.get_stats()
        P.assert_same_csr(ns.neighbor_csr(0, 0), oracle.pair_search(pts, pts, radius=r, same_set=True), f"step {step} ({mv})")
        _oracle_box(oracle, box, [pts], np.float32(1.5) * r)
        assert np.array_equal(np.array(st["world_bottom"] + st["world_top"], np.float32), box), f"world box after step {step} ({mv})"
        if mv == "first":
            assert st["speculated"] == 0
        if mv in ("escape", "blow_up"):
            assert st["speculation_redos"] == 1 and st["speculated"] == 0, f"step {step} ({mv}): leaving the box must be noticed"
            seen["redone"] += 1
        elif mv == "jitter":
            assert st["speculated"] == 1 and st["speculation_redos"] == 0 and st["ms_bounds"] == 0.0
            seen["reused"] += 1
    assert seen == {"reused": 5, "redone": 2}


def test_growing_radius_invalidates_the_grid(oracle):
    """per-point radii: the cell edge of a reused grid covers the largest radius it was built for; a larger one must be noticed"""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 40000
    pts = D.uniform_cloud(n, 5)
    r0 = D.radius_for_neighbors(n, 20.0)
    radii = (r0 * (np.float32(1.0) + np.float32(0.5) * D.uniform01(6, 0, n))).astype(np.float32)
    d_pts, d_r = torch.from_numpy(pts).cuda(), torch.from_numpy(radii).cuda()
    ns = T.TreeNSearch()
    ns.add_point_set(d_pts, d_r)
    ns.set_active_search(0, 0, True)
    for step in range(4):
        if step == 2:
            radii[777] = np.float32(2.5) * r0                      # above every radius the grid was laid out for
            d_r.copy_(torch.from_numpy(radii))
        ns.run()
        st = ns.get_stats()
        ref = oracle.pair_search(pts, pts, ra=radii, rb=radii, symmetric=True, same_set=True)
        P.assert_same_csr(ns.neighbor_csr(0, 0), ref, f"step {step}")
        assert st["speculation_redos"] == (1 if step == 2 else 0)
        assert st["speculated"] == (1 if step in (1, 3) else 0)


def test_static_set_is_cached_and_changes_are_noticed(oracle):
    """C3 shape: a moving fluid searched in itself and in a boundary that never changes.  From the third run on the boundary keeps
    its sorted arrays and cell table (n_cached_sets == 1); changing ONE coordinate of it in place -- same pointer, same size --
    is noticed through the checksum, the run is repeated and the lists reflect the change."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    fluid, boundary, r = D.two_set_cloud(40000, 10000)
    d_f, d_b = torch.from_numpy(fluid).cuda(), torch.from_numpy(boundary).cuda()
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(d_f)
    ns.add_point_set(d_b)
    ns.set_active_search(0, 0, True)
    ns.set_active_search(0, 1, True)
    rng = np.random.default_rng(1)
    for step in range(7):
        fluid += (rng.random(fluid.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(0.05) * r
        d_f.copy_(torch.from_numpy(fluid))
        if step == 4:
            boundary[4321, 1] += np.float32(0.3) * r
            d_b.copy_(torch.from_numpy(boundary))
        ns.run()
        st = ns.get_stats()
        for (i, j, a, b) in ((0, 0, fluid, fluid), (0, 1, fluid, boundary)):
            P.assert_same_csr(ns.neighbor_csr(i, j), oracle.pair_search(a, b, radius=r, same_set=(i == j)), f"step {step} pair {i}->{j}")
        if step in (2, 3, 6):
            assert st["n_cached_sets"] == 1 and st["speculation_redos"] == 0, f"step {step}: {st['n_cached_sets']} cached, {st['speculation_redos']} redos"
        if step == 4:
            assert st["speculation_redos"] == 1 and st["n_cached_sets"] == 0
        if step in (0, 1, 5):
            assert st["n_cached_sets"] == 0


def test_reuse_can_be_switched_off(oracle):
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    pts = D.uniform_cloud(20000, 9)
    r = D.radius_for_neighbors(20000, 30.0)
    ns = T.TreeNSearch(temporal_reuse=False, collect_stage_times=True)
    ns.set_search_radius(r)
    ns.add_point_set(torch.from_numpy(pts).cuda())
    ns.set_active_search(0, 0, True)
    for _ in range(3):
        ns.run()
        st = ns.get_stats()
        assert st["speculated"] == 0 and st["n_cached_sets"] == 0 and st["ms_bounds"] > 0.0
    P.assert_same_csr(ns.neighbor_csr(0, 0), oracle.pair_search(pts, pts, radius=r, same_set=True), "no reuse")


def test_zsort_between_runs_invalidates_cached_structures(oracle):
    """prepare_zsort uses the ping-pong arrays of the search structures as scratch: a cached (static) set must be rebuilt after it"""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    pts = D.uniform_cloud(30000, 13)
    r = D.radius_for_neighbors(30000, 30.0)
    d = torch.from_numpy(pts).cuda()
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(d)
    ns.set_active_search(0, 0, True)
    ref = oracle.pair_search(pts, pts, radius=r, same_set=True)
    for step in range(5):
        ns.run()
        if step == 2:
            assert ns.get_stats()["n_cached_sets"] == 1
            ns.prepare_zsort()                                     # (the order is not applied: the points stay as they are)
        if step == 3:
            assert ns.get_stats()["n_cached_sets"] == 0
        P.assert_same_csr(ns.neighbor_csr(0, 0), ref, f"step {step}")


def test_one_read_bucket_pass_and_its_overflow(oracle):
    """Round 4: in a steady-state step the bucket build reads the input ONCE -- every bucket has a window of the intermediate array sized from
    the previous run's count, tiles reserve their piece with one atomic per bucket.  (a) jittered points: the one-read pass runs, nothing is
    repeated, lists exact; (b) the same number of points inside the same box, but mirrored so that every bucket's population changes: windows
    overflow, the device raises the guard, the attempt is thrown away unseen (its sorted arrays have holes: the query kernels must not touch
    them) and repeated with the histogram pass -- lists exact; (c) the step after that reads once again."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 300000
    rng = np.random.default_rng(11)
    # a cloud that is dense at the bottom and thin at the top (z is the slowest axis of the cell key: a bucket is a few x-rows of one z-layer),
    # with fixed corner points that pin the bounding box
    # (the cloud stays clear of the faces of the box: a jitter must not pile points up on a face by clipping)
    z = rng.random(n, dtype=np.float32) ** np.float32(3.0)
    pts = (np.float32(0.05) + np.float32(0.9) * np.stack([rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32), z], axis=1)).astype(np.float32)
    pts[0] = (0.0, 0.0, 0.0)
    pts[1] = (1.0, 1.0, 1.0)
    r = np.float32(0.02)
    d = torch.from_numpy(pts).cuda()
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(d)
    ns.set_active_search(0, 0, True)

    def run_and_check(tag):
        ns.run()
        P.assert_same_csr(ns.neighbor_csr(0, 0), oracle.pair_search(pts, pts, radius=r, same_set=True), tag)
        return ns.get_stats()
    st = run_and_check("first run")
    assert st["one_read_builds"] == 0 and st["radix_passes"] == 2, "the first run has no windows yet: histogram pass"
    for k in range(2):
        pts[2:] += (rng.random((n - 2, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1) * r
        d.copy_(torch.from_numpy(pts))
        st = run_and_check(f"jitter {k}")
        assert st["one_read_builds"] == 1 and st["speculated"] == 1 and st["speculation_redos"] == 0, str({k: st[k] for k in ("one_read_builds", "speculated", "speculation_redos", "key_bits", "radix_passes", "grid_dims", "n_points")})
    pts[:, 2] = np.float32(1.0) - pts[:, 2]            # mirror in z: same box, same n, every bucket's count changes
    d.copy_(torch.from_numpy(pts))
    st = run_and_check("mirrored")
    assert st["speculation_redos"] == 1 and st["speculated"] == 0 and st["one_read_builds"] == 0, f"the overflow must be noticed and the run repeated: {st}"
    pts[2:] += (rng.random((n - 2, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1) * r
    d.copy_(torch.from_numpy(pts))
    st = run_and_check("after the repair")
    assert st["one_read_builds"] == 1 and st["speculation_redos"] == 0


def test_one_read_overflow_with_per_point_radii_user_ids_and_a_grown_set(oracle):
    """after a window overflow of the one-read bucket pass the unwritten slots of the windows hold stale points of an earlier
    run (or memory nobody ever wrote); pass B must not use their w component as an index into radii[] / ids[].  Per-point radii, user ids, a
    set that first shrinks (so that the intermediate array keeps stale rows with LARGE indices) and then grows (so that it is reallocated), and a
    mirrored cloud that overflows the windows every time: the run is repeated without speculation and the lists are exact."""
    import torch
    import treensearch_amd as T
    n_max = 260000
    rng = np.random.default_rng(23)

    def cloud(n):
        z = rng.random(n, dtype=np.float32) ** np.float32(3.0)
        p = (np.float32(0.05) + np.float32(0.9) * np.stack([rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32), z], axis=1)).astype(np.float32)
        p[0] = (0.0, 0.0, 0.0)
        p[1] = (1.0, 1.0, 1.0)
        return p
    r0 = np.float32(0.02)
    ids_all = (np.arange(n_max, dtype=np.int32)[::-1] * 2 + 5).copy()
    d_pts = torch.empty((n_max, 3), dtype=torch.float32, device="cuda")
    d_rad = torch.empty((n_max,), dtype=torch.float32, device="cuda")
    d_ids = torch.from_numpy(ids_all).cuda()
    ns = T.TreeNSearch()
    s = ns.add_point_set(d_pts, d_rad, n_points=200000)
    ns.set_active_search(s, s, True)
    ns.set_symmetric_search(True)
    ns.set_point_ids(s, d_ids)
    state = {}

    def load(n, pts, rad):
        d_pts[:n].copy_(torch.from_numpy(pts)); d_rad[:n].copy_(torch.from_numpy(rad))
        ns.resize_point_set(s, d_pts, d_rad, n_points=n)
        state.update(n=n, pts=pts, rad=rad)

    def run_and_check(tag):
        ns.run()
        n, pts, rad = state["n"], state["pts"], state["rad"]
        offs, idx = ns.neighbor_csr(s, s)
        ro, ri = oracle.pair_search(pts, pts, ra=rad, rb=rad, symmetric=True, same_set=True)
        assert np.array_equal(offs, ro), tag
        want = ids_all[ri]
        lid = np.repeat(np.arange(n), np.diff(ro))
        want = want[np.lexsort((want, lid))]
        assert np.array_equal(idx, want), tag
        return ns.get_stats()

    def jitter():
        state["pts"][2:] += (rng.random((state["n"] - 2, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1) * r0
        d_pts[:state["n"]].copy_(torch.from_numpy(state["pts"]))

    for n in (200000, 120000, n_max):          # shrink (stale rows with indices >= n stay in the intermediate array), then grow (reallocation)
        pts = cloud(n)
        rad = (r0 * (np.float32(0.6) + np.float32(0.4) * rng.random(n, dtype=np.float32))).astype(np.float32)
        rad[0] = r0                                # pins r_max, so that the grid of the first run stays valid
        load(n, pts, rad)
        run_and_check(f"n={n} first")
        jitter()
        st = run_and_check(f"n={n} jitter")
        assert st["one_read_builds"] == 1 and st["speculation_redos"] == 0, str(st)
        state["pts"][:, 2] = np.float32(1.0) - state["pts"][:, 2]      # mirror in z: every bucket's population changes, windows overflow
        d_pts[:n].copy_(torch.from_numpy(state["pts"]))
        st = run_and_check(f"n={n} mirrored")
        assert st["speculation_redos"] == 1 and st["one_read_builds"] == 0, f"the overflow must be noticed and the run repeated: {st}"
        jitter()
        st = run_and_check(f"n={n} after the repair")
        assert st["one_read_builds"] == 1 and st["speculation_redos"] == 0, str(st)


def test_one_read_bucket_pass_leaves_nan_points_out(oracle):
    """The layout of a slab's set: n owned points (they get lists), then candidates-only rows of which a varying number are real and the rest
    NaN-x padding (no points).  The one-read pass drops the NaN rows instead of letting them overflow the window of the bucket behind the last
    cell -- no repeated runs although their number changes every step -- and the lists of the owned points are exact."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n, tail = 200000, 20000
    allp = D.uniform_cloud(n + tail, 77)
    r = D.radius_for_neighbors(n, 30.0)
    d = torch.empty((n + tail, 3), dtype=torch.float32, device="cuda")
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(d)
    ns.set_query_count(0, n)
    ns.set_active_search(0, 0, True)
    rng = np.random.default_rng(5)
    for step, g in enumerate([tail, tail, 15000, 2000, 19900]):     # real rows in the tail; the others are NaN
        cur = allp.copy()
        if step:
            cur += (rng.random(cur.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(0.05) * r
            np.clip(cur, 0.0, 1.0, out=cur)
        cur[0] = (0.0, 0.0, 0.0); cur[1] = (1.0, 1.0, 1.0)
        cur[n + g:, 0] = np.nan
        cur[n + g:, 1:] = rng.random((tail - g, 2), dtype=np.float32) * np.float32(1e6)      # junk behind a NaN x counts for nothing
        d.copy_(torch.from_numpy(cur))
        ns.run()
        st = ns.get_stats()
        offs, idx = ns.neighbor_csr(0, 0)
        ro, ri = oracle.pair_search(cur[:n + g], cur[:n + g], radius=r, same_set=True)
        assert len(offs) == n + 1
        P.assert_same_csr((offs, idx), (ro[:n + 1], ri[:ro[n]]), f"step {step}")
        if step >= 2:
            assert st["one_read_builds"] == 1 and st["speculation_redos"] == 0, f"step {step}: {st}"


def test_heavy_tiers_are_launched_after_the_sync_when_needed(oracle):
    """Round 4: the two heavy tiers of a pool pass (cells with more than 512 candidates or more than 64 query points) are not launched when the
    previous run of the pair had no such cell; what the first tier passes on is counted, and when that is not zero they run after the run's
    synchronisation (tnsx_stats.heavy_catchups).  A uniform cloud (nothing heavy), then a clump appears inside the same box.  (With the
    LSD build: in the bucket build the clump overflows its bucket's window first and the whole run is repeated, all tiers included --
    the second half of the test.)"""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 120000
    pts = D.uniform_cloud(n, 9)
    pts[0] = (0.0, 0.0, 0.0); pts[1] = (1.0, 1.0, 1.0)
    r = D.radius_for_neighbors(n, 30.0)
    d = torch.from_numpy(pts).cuda()
    ns = T.TreeNSearch(bucket_build_min_points=-1)
    ns.set_search_radius(r)
    ns.add_point_set(d)
    ns.set_active_search(0, 0, True)
    for k in range(2):
        pts[7, 0] += np.float32(1e-4)                 # (an unchanged set would be taken for static: its build skipped, and repeated when it moves)
        d.copy_(torch.from_numpy(pts))
        ns.run()
        assert ns.get_stats()["heavy_catchups"] == 0
    P.assert_same_csr(ns.neighbor_csr(0, 0), oracle.pair_search(pts, pts, radius=r, same_set=True), "uniform")
    rng = np.random.default_rng(2)
    clump = np.float32(0.5) + (rng.random((3000, 3), dtype=np.float32) - np.float32(0.5)) * r      # 3000 points inside one cell edge
    pts[1000:4000] = clump
    d.copy_(torch.from_numpy(pts))
    ns.run()
    st = ns.get_stats()
    assert st["heavy_catchups"] == 1 and st["speculation_redos"] == 0, str({k: st[k] for k in ("heavy_catchups", "speculation_redos", "speculated", "pool_retries", "radix_passes", "one_read_builds")})
    P.assert_same_csr(ns.neighbor_csr(0, 0), oracle.pair_search(pts, pts, radius=r, same_set=True), "with a clump")
    ns.run()
    assert ns.get_stats()["heavy_catchups"] == 0, "the run after it launches the heavy tiers with the first"
    P.assert_same_csr(ns.neighbor_csr(0, 0), oracle.pair_search(pts, pts, radius=r, same_set=True), "with a clump, again")
    # the same scene with the default (bucket) build: the clump overflows a window of the one-read pass, the run is repeated with every tier
    pts2 = D.uniform_cloud(n, 9)
    pts2[0] = (0.0, 0.0, 0.0); pts2[1] = (1.0, 1.0, 1.0)
    d2 = torch.from_numpy(pts2).cuda()
    nb = T.TreeNSearch()
    nb.set_search_radius(r)
    nb.add_point_set(d2)
    nb.set_active_search(0, 0, True)
    for k in range(2):
        pts2[7, 0] += np.float32(1e-4)
        d2.copy_(torch.from_numpy(pts2))
        nb.run()
    assert nb.get_stats()["one_read_builds"] == 1
    pts2[1000:4000] = clump
    d2.copy_(torch.from_numpy(pts2))
    nb.run()
    st = nb.get_stats()
    assert st["speculation_redos"] == 1 and st["heavy_catchups"] == 0, st
    P.assert_same_csr(nb.neighbor_csr(0, 0), oracle.pair_search(pts2, pts2, radius=r, same_set=True), "bucket build, with a clump")


def test_first_run_of_a_large_set_sizes_its_pool_from_a_sample(oracle):
    """Round 6: the count-only pass in front of a pair's FIRST run looks at every 32nd occupied cell of a set of >= 2^20 points (tnsx_stats.sampled_passes) and the
    pool is sized from the scaled counts; the lists are what they always were (the reference's digest), a pool that falls short is repaired like any other.
    A uniform cloud (the sample is representative) and a dam break (dense column, thin floor, sparse spray: the regions differ by orders of magnitude)."""
    import torch
    import treensearch_amd as T
    import cases as CS
    from conftest import load_golden
    for name in ("uniform_fixed_2000000",):
        case = CS.by_name(name)
        ns = P.make_engine(case, 0, device_inputs=True)
        ns.run()
        st = ns.get_stats()
        assert st["sampled_passes"] == 1 and st["cold_passes"] == 0, str({k: st[k] for k in ("sampled_passes", "cold_passes", "pool_retries")})
        assert st["pool_retries"] == 0, "a uniform cloud: the scaled sample must be enough"
        P.assert_matches_golden({(0, 0): ns.neighbor_csr(0, 0, sort_each=False)}, load_golden(case.name), 0, oracle, name + " (sampled first run)", lists_sorted=False)
        ns.run()
        st = ns.get_stats()
        assert st["sampled_passes"] == 0 and st["cold_passes"] == 0 and st["pool_retries"] == 0 and st["speculated"] == 1
    from treensearch_amd import datagen as D
    n = 5_000_000
    p, rad, r0 = D.dam_break_cloud(n, 4321)
    ns = T.TreeNSearch()
    d_p, d_r = torch.from_numpy(p).cuda(), torch.from_numpy(rad).cuda()
    ns.add_point_set(d_p, d_r); ns.set_active_search(0, 0, True); ns.set_symmetric_search(True)
    ns.run()
    st = ns.get_stats()
    assert st["sampled_passes"] == 1 and st["cold_passes"] == 0 and st["pool_retries"] <= 1
    ref = T.TreeNSearch(exact_layout=True)           # the two-pass layout never sizes anything from an estimate
    ref.add_point_set(d_p, d_r); ref.set_active_search(0, 0, True); ref.set_symmetric_search(True)
    ref.run()
    assert st["n_neighbors"] == ref.get_stats()["n_neighbors"]
    a_offs, a_idx = ns.neighbor_csr_torch(0, 0, sort_each=True)
    b_offs, b_idx = ref.neighbor_csr_torch(0, 0, sort_each=True)
    assert torch.equal(a_offs, b_offs) and torch.equal(a_idx, b_idx)
