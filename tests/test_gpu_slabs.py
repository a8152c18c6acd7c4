"""configs[4] on ONE GPU: the slab path (treensearch_amd/multi.py) on the real HIP engine.

A cloud is cut into 2 / 4 / 8 x-slabs with SlabDecomposition's balanced cuts; every slab runs the SAME SlabSearch code the
multi-GPU bench runs -- tnsx_halo_pack, [owned | ghosts], candidates-only ghosts, global ids from the engine -- one thread per
emulated rank, the messages moved by an in-process transport instead of RCCL.  The union of the slabs' lists must equal the
single-device result of the engine AND the digest the real reference produced for the same cloud (tests/golden)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import cases as CS                 # noqa: E402
from conftest import load_golden   # noqa: E402
from slab_helpers import run_slabs_in_threads, union_csr   # noqa: E402

pytestmark = pytest.mark.gpu


def _engine_factory():
    import torch
    import treensearch_amd as T
    return T.TreeNSearch(stream=torch.cuda.current_stream().cuda_stream, collect_stage_times=bool(os.environ.get("TNSX_TEST_VERBOSE")))


def _single_device(case):
    """(offsets, indices unsorted) of pair 0->0 from one engine over the whole cloud"""
    import torch
    import treensearch_amd as T
    ns = T.TreeNSearch()
    variable = case.radii is not None
    if not variable:
        ns.set_search_radius(case.radius)
    pts = torch.from_numpy(case.points[0]).cuda()
    rad = torch.from_numpy(case.radii[0]).cuda() if variable else None
    ns.add_point_set(pts, rad)
    ns.set_active_search(0, 0, True)
    ns.set_symmetric_search(case.symmetric)
    ns.run()
    return ns.neighbor_csr(0, 0, sort_each=False)


def _run_slabs(case, world, n_steps=2, speculative=True, shrink_caps_before_step=None, shrink_link=None):
    import torch
    from treensearch_amd.multi import SlabDecomposition, SlabSearch
    pts_h = case.points[0]
    variable = case.radii is not None
    rad_h = case.radii[0] if variable else None
    max_r = float(rad_h.max()) if variable else float(case.radius)
    halo = max_r * 1.001
    d_all = torch.from_numpy(pts_h).cuda()
    # The torch kernels the slab step uses, once, HERE: on a fresh box the first launch of a torch kernel family pages its code in from the
    # image, which has been seen to take minutes -- inside a step that is a rank that keeps the others waiting at the transport's barrier.
    _w = torch.rand(1024, device="cuda")
    _flag = (_w.max() > 2.0) | (_w.min() > 2.0)
    _i = torch.arange(1024, device="cuda", dtype=torch.int64)
    _j = torch.empty(1024, dtype=torch.int32, device="cuda"); _j.copy_(_i)
    assert not bool(_flag.item())
    dec = SlabDecomposition(engine=_engine_factory())
    cuts = dec.balanced_cuts([d_all], plane_width=halo * 1.001, n_slabs=world)
    owner = SlabDecomposition.owner_of(d_all[:, 0], cuts).cpu().numpy()
    owned = []
    for k in range(world):
        g = np.nonzero(owner == k)[0]
        owned.append((torch.from_numpy(pts_h[g]).cuda(), torch.from_numpy(g.astype(np.int64)).cuda(),
                      torch.from_numpy(rad_h[g]).cuda() if variable else None))

    def make_slab(k, tr):
        s = SlabSearch(float(cuts[k]), float(cuts[k + 1]), None if variable else float(case.radius), _engine_factory,
                       max_radius=max_r if variable else None, transport=tr, rank=k, world=world, speculative=speculative)
        s.set_symmetric_search(case.symmetric)
        return s

    log = [[] for _ in range(world)]

    def step(k, slab, s):
        if shrink_caps_before_step == s:
            # pretend the halos were much thinner when the capacities were agreed: the speculative exchange overflows
            for p in list(slab.ex._caps):
                if shrink_link is None or {k, p} == set(shrink_link):     # (shrink_link: only the two ends of ONE link)
                    slab.ex._caps[p] = (8, 8)
        pp, gg, rr = owned[k]
        slab.step(pp, gg, rr) if variable else slab.step(pp, gg)
        log[k].append((slab.ex.speculative_last, slab.redone_last, slab.ex.rounds_last))

    slabs = run_slabs_in_threads(world, make_slab, step, n_steps)
    per_rank = []
    for k in range(world):
        offs, idx = slabs[k].engine.neighbor_csr(slabs[k].sets[0].set_id, slabs[k].sets[0].set_id, sort_each=False)
        assert len(offs) == len(owned[k][1]) + 1, "ghosts must not get lists"
        per_rank.append((owned[k][1].cpu().numpy(), offs, idx.astype(np.int64)))
    return union_csr(len(pts_h), per_rank), log, [len(o[1]) for o in owned], slabs


def _check_union(case, union, single, oracle, golden_key="0->0"):
    g_offs, g_idx = union
    s_offs, s_idx = single
    assert np.array_equal(g_offs, s_offs), "neighbour counts of the slab union differ from the single-device run"
    d_union = oracle.digest(g_offs, g_idx.astype(np.int32))
    assert d_union == oracle.digest(s_offs, s_idx), "lists of the slab union differ from the single-device run"
    fx = load_golden(case.name)["pairs"][golden_key]["strict"]
    assert int(g_offs[-1]) == fx["total"]
    assert (f"{d_union[0]:016x}", f"{d_union[1]:016x}") == (fx["digest_sum"], fx["digest_xor"]), "slab union differs from the reference's digest"
    # a few full lists, element by element
    rng = np.random.default_rng(5)
    for p in rng.integers(0, len(g_offs) - 1, 200):
        assert np.array_equal(np.sort(g_idx[g_offs[p]:g_offs[p + 1]]), np.sort(s_idx[s_offs[p]:s_offs[p + 1]]).astype(np.int64))


@pytest.fixture(scope="module")
def uniform_2m():
    case = CS.by_name("uniform_fixed_2000000")
    return case, _single_device(case)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_c5_scaled_uniform_slab_union(world, uniform_2m, oracle):
    """configs[4] at 2 M points: union of 2 / 4 / 8 slabs == single device == reference digest; the second step is speculative"""
    case, single = uniform_2m
    union, log, sizes, _ = _run_slabs(case, world)
    _check_union(case, union, single, oracle)
    assert max(sizes) < 1.2 * (len(case.points[0]) / world), f"unbalanced slabs {sizes}"
    for k in range(world):
        assert log[k][0][0] is False and log[k][0][2] == 2, "first step: exact mode, two rounds"
        assert log[k][1] == (True, False, 1), f"rank {k}: second step should be one speculative round without a redo, got {log[k][1]}"


@pytest.mark.parametrize("world", [2, 4])
def test_dam_break_slab_union_variable_radii(world, oracle):
    """clustered cloud (unequal cuts), per-point radii, symmetric search"""
    case = CS.by_name("dam_break_sym_1000000")
    single = _single_device(case)
    union, log, sizes, _ = _run_slabs(case, world)
    _check_union(case, union, single, oracle)


def test_speculative_overflow_is_redone(oracle):
    """a capacity that turns out too small: validate() fails after the run, the step is repeated in exact mode, results stay exact"""
    case = CS.by_name("uniform_fixed_1000000")
    single = _single_device(case)
    union, log, _, _ = _run_slabs(case, 4, n_steps=3, shrink_caps_before_step=1)
    _check_union(case, union, single, oracle)
    assert any(l[1][1] for l in log), "the overflow should have forced a redo"
    assert all(l[2] == (True, False, 1) for l in log), "the step after the redo is speculative again"


def test_overflow_on_one_link_is_redone_by_every_rank(oracle):
    """Only the link 1 <-> 2 of a chain of four slabs overflows: ranks 0 and 3 validate fine on their own, but the repeated step
    exchanges with BOTH neighbours, so all four must agree to repeat it (one all-reduce of a flag) -- otherwise the messages of
    ranks 1 and 2 to their healthy sides stay unmatched or pair up with the next step's."""
    case = CS.by_name("uniform_fixed_1000000")
    single = _single_device(case)
    union, log, _, _ = _run_slabs(case, 4, n_steps=4, shrink_caps_before_step=1, shrink_link=(1, 2))
    _check_union(case, union, single, oracle)
    for k in range(4):
        assert log[k][1][1] is True, f"rank {k} must take part in the repeated step"
        assert log[k][2] == (True, False, 1) and log[k][3] == (True, False, 1), f"rank {k}: the steps after the redo are speculative again"


# ----------------------------------------------------------------------------------------------------------------------
# the device-side pieces on their own
# ----------------------------------------------------------------------------------------------------------------------
def test_query_count_point_ids_and_nan_points(oracle):
    """tnsx_set_query_count (candidates-only tail), tnsx_set_point_ids (ids instead of indices), NaN x = no point"""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n, nq = 30000, 21000
    pts = D.uniform_cloud(n, 321)
    r = D.radius_for_neighbors(n, 40.0)
    absent = np.arange(n - 500, n)                       # the last 500 rows are padding
    pts_nan = pts.copy()
    pts_nan[absent, 0] = np.nan
    ids = (np.arange(n, dtype=np.int32)[::-1] * 3 + 7).copy()
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    d_pts, d_ids = torch.from_numpy(pts_nan).cuda(), torch.from_numpy(ids).cuda()
    s = ns.add_point_set(d_pts)
    ns.set_active_search(s, s, True)
    ns.set_query_count(s, nq)
    ns.set_point_ids(s, d_ids)
    for _ in range(2):
        ns.run()
    offs, idx = ns.neighbor_csr(s, s)
    ro, ri = oracle.pair_search(pts[:n - 500], pts[:n - 500], radius=r, same_set=True)
    assert len(offs) == nq + 1 and np.array_equal(offs, ro[:nq + 1])
    want = ids[ri[:ro[nq]]]
    lid = np.repeat(np.arange(nq), np.diff(ro[:nq + 1]))
    want = want[np.lexsort((want, lid))]
    assert np.array_equal(idx, want)
    # back to plain indices, every real point a query (absent points have no list: their offsets are unspecified)
    ns.set_point_ids(s, None)
    ns.set_query_count(s, n - 500)
    ns.run()
    offs, idx = ns.neighbor_csr(s, s)
    assert np.array_equal(offs, ro) and np.array_equal(idx, ri)


def test_nan_points_count_for_nothing_whatever_else_they_hold(oracle):
    """The rows of a ghost message past the real count become NaN-x points whose y, z and radius are whatever the buffer held.  They must not
    widen the bounds, raise the radius the cells are cut for (a stale radius there once made the grid of a slab 5 x too coarse: exact, and 100 x
    slower) or trip the guards of a reused grid -- in the first run (fresh bounds) and in the following ones (speculated grid) alike."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 40000
    pts = D.uniform_cloud(n, 77)
    r0 = D.radius_for_neighbors(n, 20.0)
    rng = np.random.default_rng(5)
    radii = (r0 * (1.0 + rng.random(n))).astype(np.float32)
    junk = pts.copy(); junk_r = radii.copy()
    absent = np.arange(n - 700, n)
    junk[absent, 0] = np.nan
    junk[absent, 1] = 1.0e4 * rng.standard_normal(len(absent)).astype(np.float32)     # far outside the cloud
    junk[absent[::3], 2] = np.float32(np.inf)
    junk[absent[1::3], 2] = np.float32(np.nan)
    junk_r[absent] = np.float32(50.0) * r0                                            # far above every real radius
    ref = T.TreeNSearch()
    ref.add_point_set(pts[:n - 700].copy(), radii[:n - 700].copy()); ref.set_active_search(0, 0, True); ref.set_symmetric_search(True)
    ref.run()
    want, dims = ref.neighbor_csr(0, 0), ref.get_stats()["grid_dims"]
    ns = T.TreeNSearch()
    d_pts, d_r = torch.from_numpy(junk).cuda(), torch.from_numpy(junk_r).cuda()
    s = ns.add_point_set(d_pts, d_r); ns.set_active_search(s, s, True); ns.set_symmetric_search(True)
    ns.set_query_count(s, n - 700)
    for step in range(3):
        ns.run()
        st = ns.get_stats()
        assert list(st["grid_dims"]) == list(dims), f"step {step}: the grid follows the real points only"
        assert st["speculation_redos"] == 0 and (step == 0 or st["speculated"] == 1)
        got = ns.neighbor_csr(s, s)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_translate_neighbors_kernel(oracle):
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 20000
    pts = D.uniform_cloud(n, 11)
    r = D.radius_for_neighbors(n, 30.0)
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    s = ns.add_point_set(torch.from_numpy(pts).cuda())
    ns.set_active_search(s, s, True)
    ns.run()
    id_map = np.random.default_rng(1).permutation(n).astype(np.int32) + 1000
    ns.translate_neighbors(s, s, torch.from_numpy(id_map).cuda())
    offs, idx = ns.neighbor_csr(s, s)
    ro, ri = oracle.pair_search(pts, pts, radius=r, same_set=True)
    want = id_map[ri]
    lid = np.repeat(np.arange(n), np.diff(ro))
    assert np.array_equal(offs, ro) and np.array_equal(idx, want[np.lexsort((want, lid))])


def test_x_histogram_kernel():
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    pts = D.uniform_cloud(300000, 8) * np.float32(3.0) - np.float32(1.0)
    pts[::1000, 0] = np.nan
    ns = T.TreeNSearch()
    d = torch.from_numpy(pts).cuda()
    for n_bins, x0, w in ((64, -1.0, 0.05), (5000, -0.5, 0.0004), (20000, -1.0, 0.00015)):
        h = torch.zeros(n_bins, dtype=torch.int32, device="cuda")
        inv = float(np.float32(1.0) / np.float32(w))
        ns.x_histogram(d, x0, inv, h)
        ns.synchronize()
        x = pts[:, 0][~np.isnan(pts[:, 0])]
        b = np.clip(((x - np.float32(x0)) * np.float32(inv)).astype(np.int64), 0, n_bins - 1)
        assert np.array_equal(h.cpu().numpy(), np.bincount(b, minlength=n_bins))


def test_halo_pack_asymmetric_capacities():
    """each side has its own capacity -- a big left halo with a small right buffer (and the other way round)
    must neither truncate the big side nor overrun the small one"""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 400000
    pts = D.uniform_cloud(n, 77)
    d_pts = torch.from_numpy(pts).cuda()
    gids = torch.arange(n, dtype=torch.int64, device="cuda") + (1 << 33)
    ns = T.TreeNSearch()
    counts = torch.zeros(2, dtype=torch.int32, device="cuda")
    left_cut, right_cut = 0.30, 0.995                    # ~120 k rows to the left, ~2 k to the right
    n_l, n_r = int((pts[:, 0] < np.float32(left_cut)).sum()), int((pts[:, 0] >= np.float32(right_cut)).sum())
    for cap_l, cap_r in ((n_l + 10, n_r + 3), (n_l + 10, 100), (1000, n_r + 3)):
        out_l = torch.full((cap_l, 5), -1.0, device="cuda")
        out_r = torch.full((cap_r + 1, 5), -1.0, device="cuda")           # one guard row behind the right buffer
        cl, cr = ns.halo_pack(d_pts, gids, None, left_cut, right_cut, out_l, out_r[:cap_r], counts)
        assert (cl, cr) == (n_l, n_r)
        assert torch.all(out_r[cap_r] == -1.0), "the right buffer was overrun"
        for out, cap, cnt, sel in ((out_l, cap_l, n_l, pts[:, 0] < np.float32(left_cut)), (out_r, cap_r, n_r, pts[:, 0] >= np.float32(right_cut))):
            rows = out[:min(cap, cnt)].cpu()
            got = rows[:, 3:5].contiguous().view(torch.int64).view(-1).numpy() - (1 << 33)
            assert len(np.unique(got)) == len(got) and np.all(sel[got]), "rows that do not belong to the selection"
            assert np.array_equal(rows[:, 0:3].numpy(), pts[got])
            if cnt <= cap:
                assert len(got) == cnt


# ======================================================================================================================
# The same decomposition through the C entry points of the slab layer (include/tnsx.h: tnsx_slab_balanced_cuts, tnsx_slab_create,
# tnsx_slab_step; treensearch_amd/csrc/tnsx_slab.cpp) -- what a C++ consumer calls.  The messages move through the library's
# in-process transport (tnsx_slab_transport_local: a thread per slab, one GPU); on several GPUs the same code runs over
# tnsx_slab_transport_rccl (ncclSend / ncclRecv).
# ======================================================================================================================
def _run_slabs_c(case, world, n_steps=2, speculative=True, shrink_link_before_step=None, shrink_link=None, sets=None, active=None, redistribute=False, cap_fn=None):
    """sets: [(points, radii or None)] of the WHOLE cloud (default: set 0 of the case); -> ({(i, j): union csr}, log, slabs' sizes).
    redistribute: every emulated rank starts with an equal INDEX share of every set and the points reach their owners through
    tnsx_slab_redistribute_begin / _finish (the all-to-all of the decomposition) instead of being picked on the host.
    cap_fn(step, ghosts of every rank after the previous step) -> rows or None: the agreed capacity of every link is set to that before the step."""
    import threading
    import torch
    import treensearch_amd as T
    from treensearch_amd.multi import SlabSearchC, SlabTransportC, balanced_cuts_c, redistribute_c
    if sets is None:
        sets = [(case.points[0], case.radii[0] if case.radii is not None else None)]
    active = active or [(0, 0)]
    variable = sets[0][1] is not None
    max_r = float(max(r.max() for _, r in sets)) if variable else float(case.radius)
    halo_w = max_r * 1.001
    group = SlabTransportC.local_group(world)
    d_sets = [torch.from_numpy(np.ascontiguousarray(p)).cuda() for p, _ in sets]
    # -- cuts: every emulated rank calls the collective with an equal share of the cloud (the histogram is all-reduced)
    cuts_by_rank = [None] * world
    results, errors = [None] * world, []
    barrier = threading.Barrier(world)
    owned = [None] * world
    log = [[] for _ in range(world)]
    ghosts = [0] * world

    def work(k):
        try:
            torch.cuda.set_device(0)
            eng = T.TreeNSearch()
            tr = SlabTransportC.local(group, k)
            share = [d[(d.shape[0] * k) // world:(d.shape[0] * (k + 1)) // world].contiguous() for d in d_sets]
            cuts = balanced_cuts_c(eng, tr, k, world, share, halo_w * 1.001)
            cuts_by_rank[k] = cuts
            barrier.wait()
            mine = []
            for (p_h, r_h), d in zip(sets, d_sets):
                own = np.nonzero((p_h[:, 0] >= cuts[k]) & (p_h[:, 0] < cuts[k + 1]))[0]
                if redistribute:
                    lo_i, hi_i = (len(p_h) * k) // world, (len(p_h) * (k + 1)) // world
                    g_share = torch.arange(lo_i, hi_i, dtype=torch.int64, device="cuda")
                    r_share = torch.from_numpy(np.ascontiguousarray(r_h[lo_i:hi_i])).cuda() if r_h is not None else None
                    got = redistribute_c(eng, tr, k, world, cuts, d[lo_i:hi_i].contiguous(), g_share, r_share)
                    g_own = got[1].cpu().numpy()
                    assert np.array_equal(np.sort(g_own), own), f"rank {k}: the redistribution delivered other points than the cuts assign"
                    assert np.array_equal(got[0].cpu().numpy(), p_h[g_own]) and (r_h is None or np.array_equal(got[2].cpu().numpy(), r_h[g_own])), "rows garbled on the way"
                    mine.append((got[0], got[1], got[2] if r_h is not None else None, g_own))
                    continue
                mine.append((torch.from_numpy(np.ascontiguousarray(p_h[own])).cuda(), torch.from_numpy(own.astype(np.int64)).cuda(),
                             torch.from_numpy(np.ascontiguousarray(r_h[own])).cuda() if r_h is not None else None, own))
            owned[k] = mine
            slab = SlabSearchC(float(cuts[k]), float(cuts[k + 1]), None if variable else float(case.radius), eng, tr, k, world,
                               max_radius=max_r if variable else None, speculative=speculative)
            slab.set_symmetric_search(case.symmetric)
            for (i, j) in active:
                slab.set_active_search(i, j, True)
            for s in range(n_steps):
                if cap_fn is not None and s > 0:
                    ghosts[k] = int(slab.info().n_ghost)
                    barrier.wait()
                    rows = cap_fn(s, list(ghosts))
                    barrier.wait()
                    if rows is not None:
                        for side in (0, 1):
                            slab.debug_set_capacity(side, int(rows))
                if shrink_link_before_step == s and shrink_link is not None and k in shrink_link:
                    slab.debug_set_capacity(1 if k == min(shrink_link) else 0, 8)       # both ends of ONE link
                slab.step(*[(p, g, r) if variable else (p, g) for (p, g, r, _) in mine])
                inf = slab.info()
                log[k].append((bool(inf.speculative_last), bool(inf.redone_last), int(inf.rounds_last)))
            out = {}
            for (i, j) in active:
                offs, idx = eng.neighbor_csr(slab.set_id(i), slab.set_id(j), sort_each=False)
                assert len(offs) == len(mine[i][3]) + 1, "ghosts must not get lists"
                out[(i, j)] = (mine[i][3], offs, idx.astype(np.int64))
            results[k] = out
            del slab
            tr.release()
        except BaseException as e:   # noqa: BLE001
            errors.append((k, e))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(k,)) for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    SlabTransportC.local_group_release(group)
    if errors:
        raise errors[0][1]
    for k in range(1, world):
        assert np.array_equal(cuts_by_rank[0], cuts_by_rank[k]), "every rank must arrive at the same cuts"
    unions = {pr: union_csr(len(sets[pr[0]][0]), [results[k][pr] for k in range(world)]) for pr in active}
    return unions, log, [len(owned[k][0][3]) for k in range(world)], cuts_by_rank[0]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_c_abi_slab_union_uniform(world, uniform_2m, oracle):
    """configs[4] at 2 M points through tnsx_slab_*: union of 2 / 4 / 8 slabs == single device == the reference's digest; first step
    exact (two rounds), second step ONE speculative round without a repair; the cuts equal the Python decomposition's."""
    import torch
    from treensearch_amd.multi import SlabDecomposition
    case, single = uniform_2m
    unions, log, sizes, cuts = _run_slabs_c(case, world)
    _check_union(case, unions[(0, 0)], single, oracle)
    assert max(sizes) < 1.2 * (len(case.points[0]) / world), f"unbalanced slabs {sizes}"
    for k in range(world):
        assert log[k][0] == (False, False, 2), f"rank {k}: first step should be exact with two rounds, got {log[k][0]}"
        assert log[k][1] == (True, False, 1), f"rank {k}: second step should be one speculative round without a repair, got {log[k][1]}"
    dec = SlabDecomposition(engine=_engine_factory())
    ref_cuts = dec.balanced_cuts([torch.from_numpy(case.points[0]).cuda()], plane_width=float(case.radius) * 1.001 * 1.001, n_slabs=world)
    assert np.array_equal(cuts, ref_cuts), f"{cuts} != {ref_cuts}"


def test_c_abi_overflow_on_one_link_is_repaired_by_its_two_ends(oracle):
    """Link 1 <-> 2 of four slabs overflows in a speculative step: ranks 1 and 2 move the missing rows between themselves and search
    again; ranks 0 and 3 never notice (no collective, no repeated step for them); the union is exact and the next step is
    speculative again."""
    case = CS.by_name("uniform_fixed_1000000")
    single = _single_device(case)
    unions, log, _, _ = _run_slabs_c(case, 4, n_steps=4, shrink_link_before_step=1, shrink_link=(1, 2))
    _check_union(case, unions[(0, 0)], single, oracle)
    assert log[1][1] == (True, True, 2) and log[2][1] == (True, True, 2), (log[1][1], log[2][1])
    assert log[0][1] == (True, False, 1) and log[3][1] == (True, False, 1), (log[0][1], log[3][1])
    for k in range(4):
        assert log[k][2] == (True, False, 1) and log[k][3] == (True, False, 1), f"rank {k}: the steps after the repair are speculative again"


def test_c_abi_dam_break_variable_radii(oracle):
    """clustered cloud (unequal cuts), per-point radii, symmetric search, three slabs"""
    case = CS.by_name("dam_break_sym_1000000")
    single = _single_device(case)
    unions, log, sizes, _ = _run_slabs_c(case, 3)
    _check_union(case, unions[(0, 0)], single, oracle)
    assert max(sizes) < 1.35 * (len(case.points[0]) / 3), f"unbalanced slabs {sizes}"


def test_c_abi_two_sets_asymmetric_searches(oracle):
    """two sets (fluid + boundary), searches 0->0 and 0->1 only, three slabs: both sets travel in the same exchange round"""
    case = CS.by_name("two_set_asym_800000_200000")
    sets = [(case.points[0], None), (case.points[1], None)]
    unions, log, _, _ = _run_slabs_c(case, 3, sets=sets, active=[(0, 0), (0, 1)])
    for (i, j) in [(0, 0), (0, 1)]:
        g_offs, g_idx = unions[(i, j)]
        ro, ri = oracle.pair_search(case.points[i], case.points[j], radius=case.radius, same_set=(i == j))
        assert np.array_equal(g_offs, ro)
        assert oracle.digest(g_offs, g_idx.astype(np.int32)) == oracle.digest(ro, ri, already_sorted=True)
        fx = load_golden(case.name)["pairs"][f"{i}->{j}"]["strict"]
        assert int(g_offs[-1]) == fx["total"]
    for k in range(3):
        assert log[k][1] == (True, False, 1)


def test_c_abi_capacity_edges(oracle):
    """The fixed-capacity messages of a speculative step at their edges, two slabs: capacity == the rows that travel (no padding row at all),
    capacity one row short (the link is repaired by its two ends and the step searched again), capacity several times the rows (most of the
    message is padding: NaN-x rows that must count for nothing), then exactly enough again -- every step reports what it should and the
    union after the last one is the reference's digest."""
    case = CS.by_name("uniform_fixed_1000000")
    single = _single_device(case)
    seen = {}

    def cap_fn(step, ghosts):
        # world 2: a rank's ghosts are the rows of its one link; the fuller direction decides what "exactly enough" is
        full = max(ghosts)
        seen[step] = full
        return {1: full, 2: full - 1, 3: 5 * full, 4: full}.get(step)
    unions, log, _, _ = _run_slabs_c(case, 2, n_steps=5, cap_fn=cap_fn)
    _check_union(case, unions[(0, 0)], single, oracle)
    for k in range(2):
        assert log[k][1] == (True, False, 1), f"rank {k}: capacity == rows is not an overflow: {log[k][1]}"
        assert log[k][2] == (True, True, 2), f"rank {k}: one row short: repaired by the two ends: {log[k][2]}"
        assert log[k][3] == (True, False, 1) and log[k][4] == (True, False, 1), f"rank {k}: {log[k][3:]}"


def test_c_abi_redistribute_then_search(uniform_2m, oracle):
    """the all-to-all of the decomposition behind the C ABI (tnsx_slab_redistribute_begin / _finish): four ranks that start with index shares
    of the 2 M-point cloud end up with exactly the points their slabs own, rows intact; the union of their searches is the reference's digest"""
    case, single = uniform_2m
    unions, log, sizes, _ = _run_slabs_c(case, 4, redistribute=True)
    _check_union(case, unions[(0, 0)], single, oracle)
    assert sum(sizes) == len(case.points[0])


def test_c_abi_redistribute_with_radii_and_two_sets(oracle):
    """the same with per-point radii (six-float rows, unequal cuts) in three slabs"""
    case = CS.by_name("dam_break_sym_1000000")
    single = _single_device(case)
    unions, _, sizes, _ = _run_slabs_c(case, 3, redistribute=True)
    _check_union(case, unions[(0, 0)], single, oracle)
    assert sum(sizes) == len(case.points[0])


def test_c_abi_thin_interior_slab_is_refused():
    """ghosts come from the two adjacent slabs only, so a slab with two neighbours that is thinner than the halo would silently
    lose pairs between its neighbours -- tnsx_slab_create refuses it and says why; edge slabs may be as thin as they like."""
    import ctypes as C
    import treensearch_amd as T
    from treensearch_amd.multi import SlabSearchC, SlabTransportC
    group = SlabTransportC.local_group(3)
    try:
        eng = T.TreeNSearch()
        tr = SlabTransportC.local(group, 1)
        with pytest.raises(RuntimeError, match="thinner than the halo"):
            SlabSearchC(0.50, 0.505, 0.01, eng, tr, 1, 3)
        tr.release()
        tr0 = SlabTransportC.local(group, 0)
        s0 = SlabSearchC(-np.inf, 0.001, 0.01, eng, tr0, 0, 3)     # an edge slab: fine
        del s0
        tr0.release()
    finally:
        SlabTransportC.local_group_release(group)


def test_c_abi_watchdog_names_the_link_instead_of_hanging():
    """A step whose exchange never completes (here: a transport that parks a multi-second kernel on the stream, standing in for a neighbour that
    never posts its side) returns TNSX_ERR_TIMEOUT after the watchdog's bound with the link in the message; the engine and the process live on."""
    import ctypes as C
    import time
    import torch
    import treensearch_amd as T
    from treensearch_amd import api as A
    from treensearch_amd.multi import SlabSearchC, SlabTransportC
    stalled = []
    # (the spin kernel counts ticks of a clock whose rate differs between devices: calibrate it)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.cuda._sleep(10_000_000); torch.cuda.synchronize()
    ticks_per_s = 10_000_000 / max(time.perf_counter() - t0, 1e-6)

    def exchange(user, rk, wd, ops, n_ops, stream):
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            torch.cuda._sleep(int(ticks_per_s * 4.0))      # ~4 s of spinning on the engine's stream
        stalled.append(n_ops)
        return 0
    EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(A.SlabOp), C.c_int, C.c_void_p)
    cb = EX(exchange)
    tr = SlabTransportC()
    tr.t.user, tr.t.exchange, tr.t.allreduce, tr.t.release, tr.t.abort = None, C.cast(cb, C.c_void_p), None, None, None
    tr.kind = "host"
    eng = T.TreeNSearch()
    slab = SlabSearchC(-np.inf, 0.5, 0.05, eng, tr, 0, 2)
    slab.set_watchdog(1.0)
    pts = torch.rand(20000, 3, device="cuda") * 0.5
    gid = torch.arange(20000, dtype=torch.int64, device="cuda")
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError) as ei:
        slab.step(pts, gid)
    dt = time.perf_counter() - t0
    assert "slab layer error 8" in str(ei.value) and "link 0 <-> 1" in str(ei.value) and "did not drain" in str(ei.value), str(ei.value)
    assert 0.9 < dt < 4.0, f"the watchdog should fire after ~1 s, took {dt:.2f} s"
    assert stalled, "the stand-in exchange was never called"
    torch.cuda.synchronize()                                # the parked kernel ends; nothing is left hanging
    del slab


def _mp_worker(rank, world, port, case_name, out_dir):
    """one PROCESS per rank (all on GPU 0): torch.distributed over gloo carries the slab layer's messages (SlabTransportC.host_staged)"""
    import torch
    import torch.distributed as dist
    import treensearch_amd as T
    from treensearch_amd.multi import SlabSearchC, SlabTransportC, balanced_cuts_c, redistribute_c
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    case = CS.by_name(case_name)
    p_h = case.points[0]
    n = len(p_h)
    lo_i, hi_i = (n * rank) // world, (n * (rank + 1)) // world
    eng = T.TreeNSearch()
    tr = SlabTransportC.host_staged(rank, world)
    share = torch.from_numpy(np.ascontiguousarray(p_h[lo_i:hi_i])).cuda()
    cuts = balanced_cuts_c(eng, tr, rank, world, [share], float(case.radius) * 1.002)
    pts, gids = redistribute_c(eng, tr, rank, world, cuts, share, torch.arange(lo_i, hi_i, dtype=torch.int64, device="cuda"))
    slab = SlabSearchC(float(cuts[rank]), float(cuts[rank + 1]), float(case.radius), eng, tr, rank, world)
    slab.set_watchdog(60.0)
    log = []
    for _ in range(3):
        slab.step(pts, gids)
        inf = slab.info()
        log.append((int(inf.speculative_last), int(inf.redone_last), int(inf.rounds_last)))
    offs, idx = eng.neighbor_csr(slab.set_id(0), slab.set_id(0), sort_each=False)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), gids=gids.cpu().numpy(), offs=offs, idx=idx.astype(np.int64), cuts=cuts, log=np.array(log))
    del slab
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_c_abi_across_processes_over_gloo(world, oracle, tmp_path):
    """The C entry points of the slab layer (cuts, redistribution, three steps) with one PROCESS per rank: the messages cross real process
    boundaries through an application-filled tnsx_slab_transport (torch.distributed gloo, staged through host memory), the ranks share the
    one GPU of the box.  What RCCL does on eight GPUs, minus RCCL: protocol, message sizes, all-reduces, speculation and the union."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    name = "uniform_fixed_1000000"
    mp.start_processes(_mp_worker, args=(world, port, name, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    case = CS.by_name(name)
    per_rank, cuts0 = [], None
    for k in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{k}.npz"))
        per_rank.append((d["gids"], d["offs"], d["idx"]))
        cuts0 = d["cuts"] if cuts0 is None else cuts0
        assert np.array_equal(cuts0, d["cuts"]), "every rank must arrive at the same cuts"
        log = [tuple(x) for x in d["log"]]
        assert log[0] == (0, 0, 2) and log[1] == (1, 0, 1) and log[2] == (1, 0, 1), f"rank {k}: exact step first, then single speculative rounds; got {log}"
    assert sum(len(g) for g, _, _ in per_rank) == len(case.points[0])
    g_offs, g_idx = union_csr(len(case.points[0]), per_rank)
    fx = load_golden(case.name)["pairs"]["0->0"]["strict"]
    d_union = oracle.digest(g_offs, g_idx.astype(np.int32))
    assert int(g_offs[-1]) == fx["total"]
    assert (f"{d_union[0]:016x}", f"{d_union[1]:016x}") == (fx["digest_sum"], fx["digest_xor"]), "union of the processes' lists differs from the reference's digest"


def test_rccl_transport_on_one_rank():
    """The RCCL transport of the slab layer on the only GPU there is: RCCL refuses two ranks on one device, so a world of ONE rank is all
    a single-GPU box can run -- it still goes through everything tnsx_slab_transport_rccl does (dlopen of librccl, every symbol,
    ncclGetUniqueId, ncclCommInitRank, one group of ncclSend + ncclRecv to itself on the caller's stream, ncclAllReduce of all three
    kinds, ncclCommDestroy), with the payload checked."""
    import ctypes as C
    import torch
    from treensearch_amd import api as A
    L = A.load_library()
    raw = (C.c_ubyte * 128)()
    assert L.tnsx_slab_rccl_unique_id(raw) == 0, (L.tnsx_slab_rccl_error() or b"").decode()
    tr = A.SlabTransport()
    assert L.tnsx_slab_transport_rccl(raw, 0, 1, 0, C.byref(tr)) == 0, (L.tnsx_slab_rccl_error() or b"").decode()

    class Op(C.Structure):
        _fields_ = [("peer", C.c_int), ("send", C.c_void_p), ("send_bytes", C.c_size_t), ("recv", C.c_void_p), ("recv_bytes", C.c_size_t)]
    exchange = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(Op), C.c_int, C.c_void_p)(tr.exchange)
    allreduce = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p)(tr.allreduce)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        src = torch.arange(100003, dtype=torch.float32, device="cuda") * 0.5
        dst = torch.zeros_like(src)
        op = Op(0, src.data_ptr(), src.numel() * 4, dst.data_ptr(), dst.numel() * 4)
        assert exchange(tr.user, 0, 1, C.byref(op), 1, C.c_void_p(stream.cuda_stream)) == 0
        u = torch.tensor([3, 5, 7], dtype=torch.int32, device="cuda")
        f = torch.tensor([1.5, -2.0], dtype=torch.float32, device="cuda")
        assert allreduce(tr.user, 0, 1, u.data_ptr(), 3, 0, C.c_void_p(stream.cuda_stream)) == 0      # sum of u32
        assert allreduce(tr.user, 0, 1, f.data_ptr(), 2, 1, C.c_void_p(stream.cuda_stream)) == 0      # min of f32
        assert allreduce(tr.user, 0, 1, f.data_ptr(), 2, 2, C.c_void_p(stream.cuda_stream)) == 0      # max of f32
    stream.synchronize()
    assert torch.equal(src, dst) and u.tolist() == [3, 5, 7] and f.tolist() == [1.5, -2.0]
    L.tnsx_slab_transport_release(C.byref(tr))
    assert not tr.user
