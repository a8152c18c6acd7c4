"""GPU parity tests: the HIP path through the C ABI against (a) the CPU oracle on the same seeded inputs,
(b) the golden fixtures generated from the real reference.  Bit-exact index sets (compared as per-point sorted
lists) in both arithmetic modes."""
import numpy as np
import pytest

import cases as CS
import parity as P
from conftest import load_golden

pytestmark = pytest.mark.gpu

SMALL = CS.small_cases()
MEDIUM = ["uniform_fixed_1000000", "uniform_fixed_2000000", "two_set_asym_800000_200000", "dam_break_sym_1000000"]   # (built on demand, CS.by_name)


@pytest.mark.parametrize("case", SMALL, ids=[c.name for c in SMALL])
@pytest.mark.parametrize("mode", [0, 1], ids=["strict", "contracted"])
def test_small_cases_match_oracle_and_golden(case, mode, oracle):
    res, ns = P.run_engine_case(case, mode)
    ora = P.run_oracle_case(case, mode, oracle)
    for pr in case.active:
        P.assert_same_csr(res[pr], ora[pr], f"{case.name} {pr} engine-vs-oracle")
    P.assert_matches_golden(res, load_golden(case.name), mode, oracle, case.name)
    st = ns.get_stats()
    assert st["n_neighbors"] == sum(int(ora[pr][0][-1]) for pr in case.active)


@pytest.mark.parametrize("case", MEDIUM)
@pytest.mark.parametrize("mode", [0, 1], ids=["strict", "contracted"])
def test_medium_cases_match_golden_digest(case, mode, oracle):
    case = CS.by_name(case)
    res, _ = P.run_engine_case(case, mode, device_inputs=True)
    P.assert_matches_golden(res, load_golden(case.name), mode, oracle, case.name)


@pytest.mark.parametrize("mode", [0, 1], ids=["strict", "contracted"])
def test_c2_10m_matches_reference_digest(mode, oracle):
    """BASELINE.json configs[1] at full size: 10 M uniform points.  The fixture digests come from the real
    reference's AVX2 path (592 472 324 pairs strict / 592 472 310 contracted)."""
    case = CS.by_name("uniform_fixed_10000000")
    golden = load_golden(case.name)
    ns = P.make_engine(case, mode, device_inputs=True)
    ns.run()
    st = ns.get_stats()
    g = golden["pairs"]["0->0"][P.MODE_NAMES[mode]]
    assert st["n_neighbors"] == g["total"]
    offs, idx = ns.neighbor_csr(0, 0, sort_each=False)
    P.assert_matches_golden({(0, 0): (offs, idx)}, golden, mode, oracle, case.name, lists_sorted=False)
    # size-independent property: the relation is symmetric for a fixed radius (i in N(j) <=> j in N(i))
    cnt = np.diff(offs)
    indeg = np.bincount(idx, minlength=len(cnt))
    assert np.array_equal(indeg, cnt)


def test_device_and_host_inputs_agree(oracle):
    case = CS.by_name("two_set_asym_80000_20000")
    a, _ = P.run_engine_case(case, 0, device_inputs=False)
    b, _ = P.run_engine_case(case, 0, device_inputs=True)
    for pr in case.active:
        P.assert_same_csr(a[pr], b[pr], f"host-vs-device inputs {pr}")


def test_results_do_not_depend_on_input_order(oracle):
    """Permuting the input must permute the result: lists of original point p contain original indices."""
    case = CS.by_name("uniform_fixed_100000")
    pts = case.points[0]
    perm = np.random.default_rng(1).permutation(len(pts)).astype(np.int32)
    shuffled = CS.Case("shuffled", [np.ascontiguousarray(pts[perm])], None, case.radius, [(0, 0)])
    res, _ = P.run_engine_case(shuffled, 0)
    offs_o, idx_o = oracle.remap_csr(perm, perm, *res[(0, 0)])
    ref = P.run_oracle_case(case, 0, oracle)[(0, 0)]
    P.assert_same_csr((offs_o, idx_o), ref, "shuffled input")


def test_coarsened_grid_still_exact(oracle):
    """Force the dense cell table to be tiny: the engine must coarsen its grid, never drop neighbours."""
    case = CS.by_name("uniform_fixed_100000")
    res, ns = P.run_engine_case(case, 0, max_dense_cells=512)
    assert ns.get_stats()["n_grid_cells"] <= 512
    P.assert_matches_golden(res, load_golden(case.name), 0, oracle, "coarsened")


# ------------------------------------------------------------------------------------------------------------------
# single-pass pool mode (default): every pair is built in one pass into a record pool that is sized from the previous run;
# the first run of a pair makes a dry (count-only) pass first.  exact_layout=1 selects count -> scan -> fill instead.
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", SMALL, ids=[c.name for c in SMALL])
@pytest.mark.parametrize("mode", [0, 1], ids=["strict", "contracted"])
def test_pool_mode_matches_golden(case, mode, oracle):
    ns_exact = P.make_engine(case, mode, exact_layout=True)
    ns_exact.run()                # count -> scan -> fill (general kernel)
    assert ns_exact.get_stats()["n_pool_pairs"] == 0
    exact = {pr: ns_exact.neighbor_csr(*pr) for pr in case.active}
    P.assert_matches_golden(exact, load_golden(case.name), mode, oracle, case.name + " (exact layout)")
    ns = P.make_engine(case, mode)
    n_nonempty = sum(1 for (i, j) in case.active if len(case.points[i]) > 0)
    for step in range(2):         # step 0: dry pass + sized pass, step 1: one pass
        ns.run()
        st = ns.get_stats()
        assert st["n_pool_pairs"] == n_nonempty and st["pool_retries"] == 0
        res = {pr: ns.neighbor_csr(*pr) for pr in case.active}
        for pr in case.active:
            P.assert_same_csr(res[pr], exact[pr], f"{case.name} {pr} pool (run {step}) vs exact layout")
        assert st["n_neighbors"] == sum(int(exact[pr][0][-1]) for pr in case.active)


def test_pool_overflow_is_detected_and_repaired(oracle):
    """The pool is sized from the previous run; when the point set grows it overflows, the engine notices (device
    cursor > capacity) and repeats the pass with a bigger pool.  Results stay exact."""
    import treensearch_amd as T
    case = CS.by_name("uniform_fixed_100000")
    pts = case.points[0]
    ns = T.TreeNSearch()
    ns.set_search_radius(case.radius)
    ns.add_point_set(pts, n_points=5000)
    ns.set_active_search(0, 0, True)
    ns.run()
    ns.run()
    assert ns.get_stats()["n_pool_pairs"] == 1
    ns.resize_point_set(0, pts, n_points=len(pts))      # 20x more points, ~400x more neighbours
    ns.run()
    st = ns.get_stats()
    assert st["n_pool_pairs"] == 1 and st["pool_retries"] >= 1
    P.assert_matches_golden({(0, 0): ns.neighbor_csr(0, 0)}, load_golden(case.name), 0, oracle, "after overflow")


def test_pool_regions_follow_a_moving_cluster():
    """The pool is cut into one region per XCD, each sized by what that XCD's share of the cell list produced in the previous run
    (plus a common region that takes what does not fit).  A dense cluster that jumps from one end of the domain to the other
    shifts the load from the first regions to the last ones: the lists must stay exact (common region, or a repeated pass)."""
    import treensearch_amd as T
    rng = np.random.default_rng(77)
    n_bg, n_cl = 100_000, 50_000
    bg = rng.random((n_bg, 3), dtype=np.float32) * np.array([8.0, 1.0, 1.0], np.float32)
    blob = (rng.random((n_cl, 3), dtype=np.float32) * 0.35).astype(np.float32)
    radius = 0.03
    ns = T.TreeNSearch(); ns.set_search_radius(radius)
    ref = T.TreeNSearch(exact_layout=True); ref.set_search_radius(radius)
    pts = np.ascontiguousarray(np.concatenate([bg, blob]))
    ns.add_point_set(pts); ns.set_active_search(0, 0, True)
    ref.add_point_set(pts); ref.set_active_search(0, 0, True)
    retries = 0
    for step, x0 in enumerate([0.0, 0.0, 7.6, 7.6, 0.1]):
        pts[n_bg:] = blob + np.array([x0, 0.3, 0.3], np.float32)
        ns.run(); ref.run()
        st = ns.get_stats()
        retries += st["pool_retries"]
        P.assert_same_csr(ns.neighbor_csr(0, 0), ref.neighbor_csr(0, 0), f"moving cluster, step {step} (x0 = {x0})")
        assert st["n_neighbors"] == ref.get_stats()["n_neighbors"]
    assert retries <= 3     # (a repeated pass is allowed when the cluster jumps, never in the steps where it stays)


def test_far_outliers_do_not_coarsen_the_grid(oracle):
    """A few stray points far away from a dense cloud: cells of one search radius over the bounding box do not fit any table, and
    coarser cells would make every query test thousands of candidates.  The grid is laid over the bulk of the points instead; the
    outliers are binned into its border cells, which is exact (clamping never increases a coordinate difference).  Checked against
    the CPU restatement, including outliers that are neighbours of each other and an outlier next to the cloud."""
    import treensearch_amd as T
    rng = np.random.default_rng(5)
    n = 30_000
    cloud = rng.random((n, 3), dtype=np.float32)
    radius = 0.04
    far = np.array([[400.0, 0.5, 0.5], [400.0 + 0.5 * radius, 0.5, 0.5],      # two outliers that ARE neighbours of each other
                    [-250.0, -250.0, 300.0], [0.5, 0.5, 380.0],
                    [1.0 + 0.9 * radius, 0.5, 0.5],                               # just outside the cloud: neighbour of cloud points
                    [399.0, 0.5, 0.5]], np.float32)
    pts = np.ascontiguousarray(np.concatenate([cloud, far]))
    ns = T.TreeNSearch(); ns.set_search_radius(radius)
    ns.add_point_set(pts); ns.set_active_search(0, 0, True)
    # runs 0-2: the cloud jitters in place (the trimmed grid is reused like any other); before run 3 the whole cloud moves away from
    # its grid (still exact: everything sits in border cells, but the engine notices); run 4 has a grid around the new place
    expect_reuse = [0, 1, 1, 1, 0]
    for step in range(5):
        ns.run()
        st = ns.get_stats()
        assert st["grid_trimmed"] == 1 and st["speculated"] == expect_reuse[step] and st["speculation_redos"] == 0, (step, st["speculated"])
        assert abs(st["grid_cell_size"] / radius - 1.0) < 1e-3, "the cells must stay one search radius wide"
        assert st["n_grid_cells"] < 2_000_000
        ref = oracle.pair_search(pts, pts, radius=radius, same_set=True, mode=0, use_grid=False)    # (all pairs)
        P.assert_same_csr(ns.neighbor_csr(0, 0), ref, f"cloud + far outliers, run {step}")
        pts[:n] += (rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32(0.2 * radius)
        if step == 2:
            pts[:n] += np.array([6.0, -3.0, 2.0], np.float32)
            pts[n + 4] += np.array([6.0, -3.0, 2.0], np.float32)        # (the point next to the cloud goes with it)
    off, idx = ns.neighbor_csr(0, 0)
    assert list(idx[off[n]:off[n + 1]]) == [n + 1] and list(idx[off[n + 1]:off[n + 2]]) == [n]
    # the outliers go away: the grid that is there still fits, it is reused
    ns.resize_point_set(0, pts, n_points=n)
    ns.run()
    assert ns.get_stats()["speculated"] == 1
    ref = oracle.pair_search(pts[:n], pts[:n], radius=radius, same_set=True, mode=0, use_grid=False)
    P.assert_same_csr(ns.neighbor_csr(0, 0), ref, "cloud without its outliers")
    # a fresh engine on the same points: an ordinary grid
    ns2 = T.TreeNSearch(); ns2.set_search_radius(radius)
    ns2.add_point_set(pts, n_points=n); ns2.set_active_search(0, 0, True)
    ns2.run()
    assert ns2.get_stats()["grid_trimmed"] == 0


@pytest.mark.parametrize("devices", [None, [0, 0]], ids=["one_engine", "two_engines"])
def test_far_outliers_two_sets_per_point_radii(devices, oracle):
    """the same with two sets, per-point radii (symmetric), double-precision input and all four searches -- on one engine and through
    the multi-device mode of the ABI (every slab engine trims its own grid)"""
    import treensearch_amd as T
    rng = np.random.default_rng(9)
    na, nb = 12_000, 5_000
    r0 = 0.035
    a = rng.random((na, 3))                                                   # float64 on purpose
    b = rng.random((nb, 3)).astype(np.float32) * np.float32(0.5)
    a[-3:] = [[300.0, 0.2, 0.2], [300.0 + 0.5 * r0, 0.2, 0.2], [-200.0, 250.0, 0.5]]     # (the reference allows 2^15 cells of 1.5 r_min per axis)
    b[-2:] = [[300.0, 0.2 + 0.4 * r0, 0.2], [0.25, 0.25, -250.0]]
    ra = (r0 * (1.0 + rng.random(na))).astype(np.float32)
    rb = (r0 * (1.0 + rng.random(nb))).astype(np.float32)
    ns = T.TreeNSearch(devices=devices) if devices else T.TreeNSearch()
    ns.add_point_set(a, ra.astype(np.float64)); ns.add_point_set(b, rb)      # (set 0: doubles, cast to float by the engine like TreeNSearch.cpp:277-296)
    ns.set_symmetric_search(True)
    pairs = [(0, 0), (0, 1), (1, 0), (1, 1)]
    for pr in pairs:
        ns.set_active_search(*pr, True)
    for step in range(2):
        ns.run()
        if not devices:
            assert ns.get_stats()["grid_trimmed"] == 1
        sets = [(a.astype(np.float32), ra), (b, rb)]
        for (i, j) in pairs:
            ref = oracle.pair_search(sets[i][0], sets[j][0], ra=sets[i][1], rb=sets[j][1], symmetric=True, same_set=(i == j), mode=0, use_grid=False)
            P.assert_same_csr(ns.neighbor_csr(i, j), ref, f"two sets + outliers, pair {i}->{j}, run {step}")
    off, idx = ns.neighbor_csr(0, 1)
    assert nb - 2 in idx[off[na - 3]:off[na - 2]]          # the outlier of set 0 at x = 300 sees the outlier of set 1 next to it


def test_exact_layout_option_is_sorted_and_gapless(oracle):
    case = CS.by_name("uniform_fixed_100000")
    ns = P.make_engine(case, 0, exact_layout=True)
    ns.run(); ns.run()
    st = ns.get_stats()
    assert st["n_pool_pairs"] == 0
    v = ns.pair_view(0, 0)
    assert v.n_records == v.n_neighbors + v.n_points           # no gaps
    P.assert_matches_golden({(0, 0): ns.neighbor_csr(0, 0)}, load_golden(case.name), 0, oracle, "exact layout")


def test_c2_10m_pool_mode_matches_reference_digest(oracle):
    case = CS.by_name("uniform_fixed_10000000")
    golden = load_golden(case.name)
    ns = P.make_engine(case, 0, device_inputs=True)
    ns.run(); ns.run()
    st = ns.get_stats()
    assert st["n_pool_pairs"] == 1 and st["n_neighbors"] == golden["pairs"]["0->0"]["strict"]["total"]
    P.assert_matches_golden({(0, 0): ns.neighbor_csr(0, 0, sort_each=False)}, golden, 0, oracle, case.name + " (pool)", lists_sorted=False)


def test_exact_layout_is_bitwise_reproducible():
    """exact_layout=1: stable cell sort + count/scan/fill => identical bytes from run to run."""
    case = CS.by_name("two_set_asym_80000_20000")
    ns = P.make_engine(case, 0, exact_layout=True)
    ns.run()
    a = {pr: tuple(x.copy() for x in ns.neighbor_records(*pr)) for pr in case.active}
    ns.run()
    for pr in case.active:
        b = ns.neighbor_records(*pr)
        assert np.array_equal(a[pr][0], b[0]) and np.array_equal(a[pr][1], b[1])


@pytest.mark.parametrize("max_cells", [1 << 9, 1 << 13, 1 << 17, 1 << 26])
@pytest.mark.parametrize("name", ["dam_break_sym_100000", "uniform_fixed_100000", "random_var_sym_30000_10000"])
def test_cell_sort_digit_plans(name, max_cells, oracle):
    """The build sorts the points by cell key with digits of 8..11 bits in 1..3 passes, depending on the number of grid
    cells.  max_dense_cells forces coarser grids => other key widths => other digit plans; the sets never change."""
    case = CS.by_name(name)
    ns = P.make_engine(case, 0, max_dense_cells=max_cells)
    ns.run()
    res1 = {pr: ns.neighbor_csr(*pr) for pr in case.active}
    ns.run()
    st = ns.get_stats()
    assert st["n_grid_cells"] <= max_cells
    # sets of 65536 points and more with keys of at most 24 bits: the two-pass bucket build (high-digit pass + bucket-local counting
    # sort that emits the cell table); everything else: LSD passes of 8..11 bits + k_cell_table
    n_max = max(len(p) for p in case.points)
    assert st["radix_passes"] == (2 if (n_max >= 65536 and st["key_bits"] <= 24) else (st["key_bits"] + 10) // 11)
    res2 = {pr: ns.neighbor_csr(*pr) for pr in case.active}
    P.assert_matches_golden(res1, load_golden(case.name), 0, oracle, f"{name} max_cells={max_cells} (exact pass)")
    P.assert_matches_golden(res2, load_golden(case.name), 0, oracle, f"{name} max_cells={max_cells} (pool pass)")


def test_sparse_domain_and_lazy_table_clear(oracle):
    """Two small clusters at opposite corners of a large box: > 2^26 grid cells, nearly all empty.  The dense cell table is
    gigabytes large and is never memset per run -- the next run clears exactly the entries of the previous occupied-cell
    list.  Moving the clusters (other cells, other grid dimensions) must therefore never leave stale cells behind."""
    import treensearch_amd as T
    rng = np.random.default_rng(7)
    r = np.float32(0.002)
    def cloud(shift, far):
        a = rng.random((15000, 3), dtype=np.float32) * np.float32(0.03) + np.float32(shift)
        b = rng.random((15000, 3), dtype=np.float32) * np.float32(0.03) + np.float32(far)
        return np.ascontiguousarray(np.concatenate([a, b]))
    pts = cloud(0.0, 1.2)
    ns = T.TreeNSearch(max_dense_cells=1 << 30)      # (the default bounds the table by the number of points: see the next test)
    ns.set_search_radius(r)
    ns.add_point_set(pts)
    ns.set_active_search(0, 0, True)
    ns.run()
    st = ns.get_stats()
    assert st["n_grid_cells"] > (1 << 26), st["n_grid_cells"]
    assert abs(st["grid_cell_size"] / float(r) - 1.0) < 1e-3          # not coarsened
    P.assert_same_csr(ns.neighbor_csr(0, 0), oracle.pair_search(pts, pts, radius=r, same_set=True), "sparse step 0")
    for step, (shift, far) in enumerate([(0.011, 1.2), (0.0, 0.9), (0.3, 0.35), (0.0, 1.2)]):
        pts[:] = cloud(shift, far)
        ns.run()
        P.assert_same_csr(ns.neighbor_csr(0, 0), oracle.pair_search(pts, pts, radius=r, same_set=True), f"sparse step {step + 1}")
        assert ns.get_stats()["n_pool_pairs"] == 1


# ------------------------------------------------------------------------------------------------------------------
# size-independent properties at full size (no oracle, no fixture): they hold for ANY correct neighbour search
# ------------------------------------------------------------------------------------------------------------------
def _pair_moments(offs, idx):
    """Two order-independent 64-bit moments of the directed pair set {(i, j)}: sum a_i * b_j and sum b_i * a_j (wrapping)."""
    n = len(offs) - 1
    rng = np.random.default_rng(99)
    a = rng.integers(1, 1 << 62, n, dtype=np.uint64)
    b = rng.integers(1, 1 << 62, n, dtype=np.uint64)
    counts = np.diff(offs).astype(np.int64)
    with np.errstate(over="ignore"):
        sb = np.add.reduceat(b[idx], offs[:-1][counts > 0]) if len(idx) else np.zeros(0, np.uint64)   # sum of b_j over the list of i
        sa = np.add.reduceat(a[idx], offs[:-1][counts > 0]) if len(idx) else np.zeros(0, np.uint64)
        nz = counts > 0
        return np.sum(a[nz] * sb, dtype=np.uint64), np.sum(b[nz] * sa, dtype=np.uint64)


@pytest.mark.parametrize("n", [1_000_000])   # (the 10 M cloud is checked against the reference itself above, 20 M / 50 M / 200 M clouds through the same properties on the
                                            #  device in tests/test_gpu_fullsize.py; a 4 M instance of this host-side check took 117 s of the suite's budget)
def test_full_size_properties_symmetry_idempotence_self_exclusion(n):
    """Fixed radius, one set: (i, j) is a pair <=> (j, i) is (the fp32 predicate is symmetric under negation of the
    difference vector); no point is its own neighbour; list entries are distinct; a second run() returns the same sets."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    pts = torch.from_numpy(D.uniform_cloud(n, 2024)).cuda()
    ns = T.TreeNSearch()
    ns.set_search_radius(D.radius_for_neighbors(n))
    ns.add_point_set(pts)
    ns.set_active_search(0, 0, True)
    ns.run()
    offs, idx = ns.neighbor_csr(0, 0)                       # lists ascending
    assert offs[-1] == len(idx) and len(offs) == n + 1
    m_ij, m_ji = _pair_moments(offs, idx)
    assert m_ij == m_ji, "pair set is not symmetric"
    owner = np.repeat(np.arange(n, dtype=idx.dtype), np.diff(offs))
    assert not np.any(idx == owner), "a point lists itself"
    inner = np.ones(len(idx), bool); inner[offs[1:-1][offs[1:-1] < len(idx)]] = False
    assert np.all(np.diff(idx.astype(np.int64))[inner[1:]] > 0), "duplicate or unsorted entries inside a list"
    ns.run()
    offs2, idx2 = ns.neighbor_csr(0, 0)
    assert np.array_equal(offs, offs2) and np.array_equal(idx, idx2), "second run differs"


def _filament(n, turns, seed, jitter):
    """n points along a helix that winds through the unit cube, jittered: sparse EVERYWHERE -- a bounding box of ~10^9 cells of one radius, < 0.1 % occupied"""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.random(n))
    ang = 2.0 * np.pi * turns * t
    p = np.stack([0.5 + 0.4 * np.cos(ang), 0.5 + 0.4 * np.sin(ang), 0.05 + 0.9 * t], axis=1)
    p += (rng.random((n, 3)) - 0.5) * jitter
    return np.ascontiguousarray(p.astype(np.float32))


def test_sparse_grid_keeps_cells_of_one_radius(oracle):
    """Round 4: a cloud that is sparse everywhere (a filament through the whole box) used to get coarser cells until a dense table fitted.  Now the
    cells keep their edge of one search radius and the grid is held as key-ordered lists of occupied cells with a block index (tnsx_stats.grid_sparse);
    lists exact against the CPU restatement, over moving points (the sparse grid is reused like any other), with per-point radii, and for a pair of
    two different sets."""
    import treensearch_amd as T
    n = 300000
    r = np.float32(0.0009)
    pts = _filament(n, 14.0, 3, 0.5 * float(r))
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(pts)
    ns.set_active_search(0, 0, True)
    rng = np.random.default_rng(1)
    for step in range(3):
        ns.run()
        st = ns.get_stats()
        assert st["grid_sparse"] == 1 and st["n_grid_cells"] > (1 << 29), st["n_grid_cells"]
        assert abs(st["grid_cell_size"] / float(r) - 1.0) < 1e-3, "the cells must stay one search radius wide"
        assert st["n_occupied_cells"] < st["n_grid_cells"] // 1000
        assert step == 0 or (st["speculated"] == 1 and st["speculation_redos"] == 0)
        ro, ri = oracle.pair_search(pts, pts, radius=r, same_set=True)
        assert ro[-1] > 10 * n, "the filament should have neighbours to find"
        P.assert_same_csr(ns.neighbor_csr(0, 0), (ro, ri), f"filament, run {step}")
        pts[2:] += (rng.random((n - 2, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1) * r
    # the same grid forced dense-and-coarse (rounds 1-3): identical lists
    old = T.TreeNSearch(sparse_grid=-1)
    old.set_search_radius(r); old.add_point_set(pts); old.set_active_search(0, 0, True); old.run()
    assert old.get_stats()["grid_sparse"] == 0 and old.get_stats()["grid_cell_size"] > 1.5 * float(r)
    ns.run()
    a, b = ns.neighbor_csr(0, 0), old.neighbor_csr(0, 0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # two sets (filament + a second, shifted one), searches 0->0, 0->1 and 1->0, per-point radii, symmetric
    q = _filament(120000, 14.0, 4, 0.5 * float(r)) + np.float32(0.0004)
    ra = (r * (1.0 + rng.random(n))).astype(np.float32)
    rq = (r * (1.0 + rng.random(len(q)))).astype(np.float32)
    nv = T.TreeNSearch(sparse_grid=1)     # (cells of 2 r: the grid is within 8 x of the dense bound, where the default coarsens instead)
    nv.add_point_set(pts, ra); nv.add_point_set(q, rq)
    for (i, j) in [(0, 0), (0, 1), (1, 0)]:
        nv.set_active_search(i, j, True)
    nv.set_symmetric_search(True)
    for step in range(2):
        nv.run()
        assert nv.get_stats()["grid_sparse"] == 1
        sets = [(pts, ra), (q, rq)]
        for (i, j) in [(0, 0), (0, 1), (1, 0)]:
            ref = oracle.pair_search(sets[i][0], sets[j][0], ra=sets[i][1], rb=sets[j][1], symmetric=True, same_set=(i == j))
            P.assert_same_csr(nv.neighbor_csr(i, j), ref, f"two filaments, per-point radii, pair {i}->{j}, run {step}")


def test_an_unavailable_formulation_falls_back_to_the_cell_kernels(oracle):
    """The product library (built without the group formulation: a refutation that lives in tools/ since round 6) accepts query_formulation = 1 and runs the cell kernels: same lists, no
    group pair in the statistics."""
    import treensearch_amd as T
    assert T.load_library().tnsx_query_formulation_available(1) == 0, "the product library does not carry the refuted formulation (tools/build_group_variant.sh builds a variant that does)"
    case = CS.by_name("uniform_fixed_100000")
    ns = P.make_engine(case, 0, query_formulation=1)
    ns.run(); ns.run()
    assert ns.get_stats()["n_group_pairs"] == 0
    P.assert_matches_golden({(0, 0): ns.neighbor_csr(0, 0)}, load_golden(case.name), 0, oracle, case.name + " (formulation 1 asked of a library without it)")


@pytest.mark.parametrize("mode", [0, 1], ids=["strict", "contracted"])
def test_register_cull_keeps_candidates_on_the_radius(mode, oracle):
    """Round 6: cells of 4..7 candidate chunks are culled against the bounding box of their query points before the query loop (tnsx_query.hip,
    cull_cell_to_stage); the bound is a lower bound of the squared distance IN THE PREDICATE'S OWN ARITHMETIC, so a candidate at exactly d == r of a query
    must survive it.  A lattice of spacing r / 2 (eight points per cell: 216 candidates per cell, four chunks -- the path is taken), moved off the origin so
    that the coordinates round: every point has six neighbours at exactly d == r and many more one ulp either side."""
    import treensearch_amd as T
    r = np.float32(0.0625)
    g = np.arange(32, dtype=np.float32) * (r / np.float32(2))
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + np.float32([0.7, 1.3, 2.9])
    pts = np.ascontiguousarray(pts.astype(np.float32))
    rng = np.random.default_rng(3)
    pts = pts[rng.permutation(len(pts))]
    ns = T.TreeNSearch(arith=mode)
    ns.set_search_radius(float(r)); ns.add_point_set(pts); ns.set_active_search(0, 0, True)
    want = oracle.pair_search(pts, pts, radius=float(r), same_set=True, mode=mode)
    assert int(want[0][-1]) >= 30 * len(pts) * 0.8          # (32 neighbours within r on the lattice, fewer at the faces)
    for step in range(2):
        ns.run()
        P.assert_same_csr(ns.neighbor_csr(0, 0), want, f"lattice of spacing r / 2, run {step}")
    st = ns.get_stats()
    assert st["n_occupied_cells"] > 0 and len(pts) / st["n_occupied_cells"] >= 6.0, "the cells must be dense enough for the culled path (>= 4 chunks of candidates)"
