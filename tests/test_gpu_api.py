"""GPU tests of the drop-in API semantics.  Each test mirrors a scenario of the reference's own test program
(/root/reference/tests/tests.cpp) or a documented behaviour of tns::TreeNSearch, driven through the C ABI."""
import os
import subprocess

import numpy as np
import pytest

import cases as CS
from conftest import load_golden
import parity as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_pair(orc, pts_i, pts_j, same, radius=None, ri=None, rj=None, symmetric=True):
    if ri is None:
        return orc.pair_search(pts_i, pts_j, radius=radius, same_set=same, mode=0)
    return orc.pair_search(pts_i, pts_j, ra=ri, rb=rj, symmetric=symmetric, same_set=same, mode=0)


@pytest.mark.parametrize("n_points", [1, 100, 10000])
def test_one_set_fixed_radius_with_zsort_roundtrip(n_points, oracle):
    """tests.cpp:91-112 + the common body :34-48: run, compare with brute force, zsort, apply, run again, compare."""
    import treensearch_amd as T
    case = CS.one_set_fixed_radius(n_points)
    pts = case.points[0].copy()
    ns = T.TreeNSearch()
    ns.set_search_radius(case.radius)
    s = ns.add_point_set(pts)
    ns.set_active_search(s, s, True)
    ns.run()
    P.assert_same_csr(ns.neighbor_csr(s, s), _oracle_pair(oracle, pts, pts, True, radius=case.radius), "before zsort")
    ns.prepare_zsort()
    order = ns.get_zsort_order(s)
    assert sorted(order.tolist()) == list(range(len(pts)))
    before = pts.copy()
    ns.apply_zsort(s, pts, 3)            # permutes the user array in place; the engine re-reads it at run()
    assert np.array_equal(pts, before[order])
    ns.run()
    P.assert_same_csr(ns.neighbor_csr(s, s), _oracle_pair(oracle, pts, pts, True, radius=case.radius), "after zsort")
    # the order is Morton ordered w.r.t. the reference grid (cell-level key on the world box)
    st = ns.get_stats()
    cell = np.float32(1.5) * np.float32(case.radius)
    keys = oracle.zsort_keys(before, np.array(st["world_bottom"], np.float32), np.float32(1.0) / cell)
    assert oracle.check_zsort(keys, order) == 0


@pytest.mark.parametrize("n_points", [100, 10000])
def test_two_dynamic_sets_variable_radius(n_points, oracle):
    """tests.cpp:114-145."""
    case = CS.two_sets_variable_radius(n_points)
    res, ns = P.run_engine_case(case, 0)
    for (i, j) in case.active:
        ref = _oracle_pair(oracle, case.points[i], case.points[j], i == j, ri=case.radii[i], rj=case.radii[j])
        P.assert_same_csr(res[(i, j)], ref, f"{i}->{j}")
    assert not ns.is_search_active(1, 1)
    with pytest.raises(Exception):
        ns.neighbor_csr(1, 1)          # inactive pair (undefined behaviour in the reference, an error here)


def test_mixed_float_double_point_sets(oracle):
    """tests.cpp:147-186: one set float, the other double; doubles are cast with (float) on the device."""
    case = CS.mixed_float_double(10000)
    assert case.points[1].dtype == np.float64
    res, _ = P.run_engine_case(case, 0)
    ora = P.run_oracle_case(case, 0, oracle)
    for pr in case.active:
        P.assert_same_csr(res[pr], ora[pr], f"{pr}")


def test_resize_variable_radius(oracle):
    """tests.cpp:188-237: n/2 -> n -> n/3 through resize_point_set."""
    import treensearch_amd as T
    c = CS.two_sets_variable_radius(10000)
    p0, p1, r0, r1 = c.points[0], c.points[1], c.radii[0], c.radii[1]
    ns = T.TreeNSearch()
    ns.add_point_set(p0, r0, n_points=len(p0) // 2)
    ns.add_point_set(p1, r1, n_points=len(p1) // 2)
    for (i, j) in c.active:
        ns.set_active_search(i, j, True)

    def check(n0, n1, what):
        ns.run()
        sub = [p0[:n0], p1[:n1]]
        rad = [r0[:n0], r1[:n1]]
        for (i, j) in c.active:
            ref = _oracle_pair(oracle, sub[i], sub[j], i == j, ri=rad[i], rj=rad[j])
            P.assert_same_csr(ns.neighbor_csr(i, j), ref, f"{what} {i}->{j}")

    check(len(p0) // 2, len(p1) // 2, "original")
    ns.resize_point_set(0, p0, r0, n_points=len(p0))
    ns.resize_point_set(1, p1, r1, n_points=len(p1))
    check(len(p0), len(p1), "resize x2")
    ns.resize_point_set(0, p0, r0, n_points=len(p0) // 3)
    ns.resize_point_set(1, p1, r1, n_points=len(p1) // 3)
    check(len(p0) // 3, len(p1) // 3, "resize x0.33")


def test_moving_points_are_reread_every_run(oracle):
    """The library keeps the user's pointer and re-reads it at every run() (TreeNSearch.h:375-378)."""
    import treensearch_amd as T
    pts = CS.uniform_fixed(100000).points[0][:20000].copy()
    r = np.float32(0.06)
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(pts)
    ns.set_active_search(0, 0, True)
    ns.run()
    P.assert_same_csr(ns.neighbor_csr(0, 0), oracle.pair_search(pts, pts, radius=r, same_set=True), "step 0")
    pts += (np.random.default_rng(0).random(pts.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(0.01)
    ns.run()
    P.assert_same_csr(ns.neighbor_csr(0, 0), oracle.pair_search(pts, pts, radius=r, same_set=True), "step 1")


def test_defaults_and_getters():
    """All searches inactive by default (TreeNSearch.cpp:357-361); set_active_search(int,bool,bool) writes the
    column first, then the row (TreeNSearch.cpp:223-232)."""
    import treensearch_amd as T
    ns = T.TreeNSearch()
    ns.set_search_radius(0.1)
    a = np.zeros((4, 3), np.float32)
    for _ in range(3):
        ns.add_point_set(a)
    assert ns.get_n_sets() == 3 and ns.get_total_n_points() == 12 and ns.get_n_points_in_set(1) == 4
    assert not any(ns.is_search_active(i, j) for i in range(3) for j in range(3))
    ns.set_active_search(1, True, False)     # search in all, be found by none
    assert [ns.is_search_active(1, j) for j in range(3)] == [True, True, True]
    assert [ns.is_search_active(j, 1) for j in range(3)] == [False, True, False]
    ns.set_active_search(1, False, True)
    assert [ns.is_search_active(1, j) for j in range(3)] == [False, False, False]
    assert [ns.is_search_active(j, 1) for j in range(3)] == [True, False, True]
    ns.set_all_searches(True)
    assert all(ns.is_search_active(i, j) for i in range(3) for j in range(3))
    ns.set_active_search(0, 2, False)
    assert not ns.is_search_active(0, 2)
    assert ns.does_set_exist(2) and not ns.does_set_exist(3)


def test_error_behaviour_matches_reference_messages():
    import treensearch_amd as T
    ns = T.TreeNSearch()
    a = np.zeros((4, 3), np.float32)
    ns.add_point_set(a, np.ones(4, np.float32))
    with pytest.raises(T.TnsxError, match="Cannot set a global search radius"):     # TreeNSearch.cpp:22-25
        ns.set_search_radius(0.1)
    ns.set_cell_size(0.5)
    with pytest.raises(T.TnsxError, match="Cell size already set"):                  # TreeNSearch.cpp:175-178
        ns.set_cell_size(0.7)
    ns2 = T.TreeNSearch()
    ns2.add_point_set(a)                         # fixed-radius set but no radius given, plus a variable set
    ns2.add_point_set(a, np.ones(4, np.float32))
    with pytest.raises(T.TnsxError, match="not all point sets have per-point search radius"):   # :388-391
        ns2.run()
    ns3 = T.TreeNSearch()
    ns3.set_search_radius(0.001)
    ns3.add_point_set(np.array([[0, 0, 0], [100, 0, 0]], np.float32))
    ns3.set_active_search(0, 0)
    with pytest.raises(T.TnsxError, match="Max allowed cells per dimension is 32768"):          # :510-515
        ns3.run()
    with pytest.raises(T.TnsxError):
        ns3.resize_point_set(5, a)               # TreeNSearch.cpp:69-72


def test_get_neighborlist_and_for_each_neighbor(oracle):
    case = CS.one_set_fixed_radius(100)
    _, ns = P.run_engine_case(case, 0, mirror_to_host=True)
    offs, idx = P.run_oracle_case(case, 0, oracle)[(0, 0)]
    v = ns.pair_view(0, 0)
    assert v.records_host and v.offsets_host and v.n_neighbors == offs[-1]
    for p in (0, 7, len(offs) - 2):
        nl = ns.get_neighborlist(0, 0, p)
        assert nl.size() == offs[p + 1] - offs[p]
        assert sorted(nl[k] for k in range(nl.size())) == idx[offs[p]:offs[p + 1]].tolist()
        got = []
        ns.for_each_neighbor(0, 0, p, got.append)
        assert sorted(got) == idx[offs[p]:offs[p + 1]].tolist()
    # round 5: the host mirror is a gap-free copy in POINT order (compacted on the device before it crosses the link): the record of point p + 1 starts right
    # behind the record of point p, whatever the layout of the pool on the device
    import ctypes as C
    n = len(offs) - 1
    ho = np.ctypeslib.as_array(C.cast(v.offsets_host, C.POINTER(C.c_uint64)), shape=(n,)).copy()
    hr = np.ctypeslib.as_array(C.cast(v.records_host, C.POINTER(C.c_int32)), shape=(int(offs[-1]) + n,)).copy()
    assert ho[0] == 0 and np.array_equal(np.diff(ho), np.diff(offs)[:-1] + 1), "host records must follow each other in point order without gaps"
    assert np.array_equal(hr[ho], np.diff(offs)), "count words of the mirrored records"
    # memory in use by the lists (TreeNSearch.cpp:254-261 sums its chunk storage): the gap-free layout is exactly one count
    # word + the indices per point, the default record pool may add unused slab tails
    assert ns.get_neighborlist_n_bytes() >= 4 * (int(offs[-1]) + len(offs) - 1)
    _, ns_exact = P.run_engine_case(case, 0, exact_layout=True)
    assert ns_exact.get_neighborlist_n_bytes() == 4 * (int(offs[-1]) + len(offs) - 1)


def test_apply_zsort_on_device_and_other_dtypes(oracle):
    import torch
    import treensearch_amd as T
    case = CS.dam_break(100000)
    pts, rad = case.points[0].copy(), case.radii[0].copy()
    ns = T.TreeNSearch()
    ns.add_point_set(pts, rad)
    ns.set_active_search(0, 0)
    ns.prepare_zsort()
    order = ns.get_zsort_order(0)
    ids = np.arange(len(pts), dtype=np.int64)
    vel = np.random.default_rng(0).random((len(pts), 3)).astype(np.float64)
    d_pts = torch.from_numpy(pts.copy()).cuda()
    d_u8 = torch.arange(len(pts), dtype=torch.int64).cuda().to(torch.uint8)
    ns.apply_zsort(0, ids)
    ns.apply_zsort(0, vel, 3)
    ns.apply_zsort(0, d_pts, 3)
    ns.apply_zsort(0, d_u8, 1)
    assert np.array_equal(ids, order)
    assert np.array_equal(vel, np.random.default_rng(0).random((len(pts), 3)).astype(np.float64)[order])
    assert np.array_equal(d_pts.cpu().numpy(), pts[order])
    assert np.array_equal(d_u8.cpu().numpy(), (np.arange(len(pts)) % 256).astype(np.uint8)[order])


def test_cpp_dropin_scenarios(built_library, tmp_path):
    """The C++ shim (`#include <TreeNSearch>`) run through the reference's test scenarios, compiled with g++."""
    exe = tmp_path / "shim_scenarios"
    lib_dir = os.path.dirname(built_library)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_scenarios.cpp"), "-o", str(exe),
                           "-L" + lib_dir, "-ltnsx", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "FAILED" not in out.stdout and "ALL PASSED" in out.stdout


def test_cpp_dropin_stress(built_library, tmp_path):
    """The reference's stress scenarios through the C++ shim, every configuration compared with an all-pairs search:
    dynamic emitter (tests.cpp:434-514: two sets that start empty with null pointers, 400 random grow / shrink / replace steps)
    and the size lattice (tests.cpp:287-427: 1-3 sets, all combinations of awkward sizes, run -> zsort -> run)."""
    exe = tmp_path / "shim_stress"
    lib_dir = os.path.dirname(built_library)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_stress.cpp"), "-o", str(exe),
                           "-L" + lib_dir, "-ltnsx", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([str(exe), "400", "1"], capture_output=True, text=True, timeout=1500)
    print(out.stdout[-3000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "FAILED" not in out.stdout and "ALL PASSED" in out.stdout


def test_halo_pack_kernel_matches_torch_selection():
    """tnsx_halo_pack (multi-GPU ghost selection): same rows as the torch compare / nonzero / index_select chain, as sets;
    too small buffers report the needed size instead of overflowing."""
    import torch
    import treensearch_amd as T
    from treensearch_amd.multi import SlabExchange
    n = 300_000
    g = torch.Generator().manual_seed(5)
    pts = torch.rand((n, 3), generator=g).cuda()
    pts[:, 0] += 3.0                                             # slab [3, 4)
    radii = (torch.rand(n, generator=g) * 0.01 + 0.01).cuda()
    gids = (torch.arange(n, dtype=torch.int64) * 7919 + (1 << 40)).cuda()   # ids that need all 64 bits
    ns = T.TreeNSearch()
    for r in (None, radii):
        cols = 5 if r is None else 6
        lcut, rcut = 3.0 + 0.02, 4.0 - 0.02
        ex = SlabExchange(3.0, 4.0, 0.02, packer=ns)
        sl, sr = ex._pack_device(pts, gids, r, True, True, cols - 1)
        for got, mask in ((sl[1:], pts[:, 0] < lcut), (sr[1:], pts[:, 0] >= rcut)):   # row 0 is the message header
            sel = torch.nonzero(mask).squeeze(1)
            assert got.shape == (sel.numel(), cols)
            got_ids = got[:, cols - 2:cols].contiguous().view(torch.int64).view(-1)
            order = torch.argsort(got_ids)
            assert torch.equal(got_ids[order], gids[sel])         # gids ascend with the index
            assert torch.equal(got[order, 0:3], pts[sel])
            if r is not None:
                assert torch.equal(got[order, 3], r[sel])
        # one side only, and a buffer that is too small at first (the exchange grows it and repeats)
        ex2 = SlabExchange(3.0, 4.0, 0.25, packer=ns)
        ex2._send_buf = [None, torch.empty((16, cols), dtype=torch.float32, device="cuda")]
        l2, r2 = ex2._pack_device(pts, gids, r, False, True, cols - 1)
        assert l2 is None and r2.shape[0] - 1 == int((pts[:, 0] >= 4.0 - 0.25).sum())


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 100003])
@pytest.mark.parametrize("on_device", [False, True])
def test_grid_and_world_box_follow_the_tight_bounds(n, on_device, oracle):
    """Binning clamps coordinates, so a wrong bounding box would not change a single neighbour list -- only the speed.  The box
    geometry is therefore checked on its own, from the exact min / max of the points (extreme points placed at the ends of the
    array and in odd positions, sizes around the 4-point vector width): the world box is the reference's snapping of the tight
    bounds (TreeNSearch.cpp:474-521, restated in the oracle), and the search grid covers the tight bounds widened by two radii,
    clipped to the world box, with floor(extent / h) + 1 cells per axis."""
    import torch
    import treensearch_amd as T
    rng = np.random.default_rng(n)
    pts = (rng.random((n, 3), dtype=np.float32) * np.array([0.7, 0.2, 1.3], np.float32) + np.array([-3.0, 5.0, 0.1], np.float32)).astype(np.float32)
    pts[n - 1] += np.float32(0.4)          # maxima at the very end
    pts[(n - 1) // 2, 1] -= np.float32(0.3)
    pts = np.ascontiguousarray(pts)
    r = np.float32(0.013)
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(torch.from_numpy(pts).cuda() if on_device else pts)
    ns.set_active_search(0, 0, True)
    ns.run()
    st = ns.get_stats()
    tight = np.concatenate([pts.min(axis=0), pts.max(axis=0)]).astype(np.float32)
    box = np.array([np.finfo(np.float32).max] * 3 + [-np.finfo(np.float32).max] * 3, np.float32)    # the empty box every engine starts with
    rc, n_pow2 = oracle.world_box_update(box, oracle.tight_bounds(pts, simd=True), np.float32(1.5) * r)   # (run(): tight box + origin)
    assert rc == 1                                           # (1: the box was replaced)
    assert np.array_equal(np.array(st["world_bottom"] + st["world_top"], np.float32), box) and st["world_cells_pow2"] == n_pow2
    h = float(st["grid_cell_size"])
    lo = np.maximum(tight[:3] - np.float32(2.0) * r, box[:3])
    hi = np.minimum(tight[3:] + np.float32(2.0) * r, box[3:])
    assert np.array_equal(np.array(st["grid_origin"], np.float32), lo)
    expect = [int(np.floor((float(hi[d]) - float(lo[d])) / h)) + 1 for d in range(3)]
    assert list(st["grid_dims"]) == expect


@pytest.mark.parametrize("name", ["uniform_fixed_100000", "random_var_asym_12000_6000_ratio10", "dense_blob"])
def test_sorted_lists_equal_the_reference_without_sorting(name, oracle):
    """tnsx_options.sorted_lists: the records are ascending as they come out of the engine -- compared with the
    reference-checked lists element by element with NO sort on either side (what BruteforceNSearch.cpp:129-137 needs a sort for).
    Covers all three tiers of the record sort: <= 64 entries (lanes), <= 2048 (LDS), longer (in place in global memory)."""
    import treensearch_amd as T
    if name == "dense_blob":
        from treensearch_amd import datagen as D
        blob = D.uniform_cloud(2600, 4) * np.float32(0.004) + np.float32(0.5)          # 2600 points inside one search sphere
        rest = D.uniform_cloud(20000, 5)
        case = CS.Case("dense_blob", [np.ascontiguousarray(np.concatenate([rest[:10000], blob, rest[10000:]]))], None, np.float32(0.02), [(0, 0)])
    else:
        case = CS.by_name(name)
    ref = P.run_oracle_case(case, 0, oracle)
    ns = P.make_engine(case, 0, device_inputs=True, sorted_lists=True, collect_stage_times=True)
    for _ in range(2):
        ns.run()
    assert ns.get_stats()["ms_sort_lists"] > 0.0
    longest = 0
    for pr in case.active:
        offs, idx = ns.neighbor_csr(*pr, sort_each=False)
        assert np.array_equal(offs, ref[pr][0]) and np.array_equal(idx, ref[pr][1]), f"{name} pair {pr}: records are not the ascending lists"
        longest = max(longest, int(np.diff(offs).max()) if len(offs) > 1 else 0)
    assert longest > {"uniform_fixed_100000": 64, "random_var_asym_12000_6000_ratio10": 200, "dense_blob": 2048}[name], longest
    # and the default leaves the order alone (same sets)
    plain = P.make_engine(case, 0, device_inputs=True)
    plain.run()
    for pr in case.active:
        P.assert_same_csr(plain.neighbor_csr(*pr), ref[pr], f"{name} {pr} unsorted engine, sorted for the comparison")


def test_cpp_shim_inactive_pair_is_an_error_message(built_library, tmp_path):
    """get_neighborlist for a pair that was not active at the last run(): message on stdout + exit(-1), as INTEGRATION.md says
    (the reference asserts in debug builds and reads through a null pointer otherwise, TreeNSearch.cpp:243-246)"""
    src = tmp_path / "inactive.cpp"
    src.write_text("""
#include <TreeNSearch>
#include <cstdio>
#include <vector>
int main() {
    std::vector<float> a = {0.f, 0.f, 0.f, 0.1f, 0.f, 0.f}, b = {0.05f, 0.f, 0.f};
    tns::TreeNSearch ns;
    ns.set_search_radius(0.2f);
    ns.add_point_set(a.data(), 2);
    ns.add_point_set(b.data(), 1);
    ns.set_active_search(0, 1, true);
    ns.run();
    std::printf("active %d\\n", ns.get_neighborlist(0, 1, 0).size());
    std::fflush(stdout);
    return ns.get_neighborlist(1, 0, 0).size();      // never activated
}
""")
    exe = tmp_path / "inactive"
    lib_dir = os.path.dirname(built_library)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fopenmp", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L" + lib_dir, "-ltnsx", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert "active 1" in out.stdout
    assert out.returncode == 255 and "not active at the last run" in out.stdout, (out.returncode, out.stdout, out.stderr)


@pytest.mark.parametrize("name", ["uniform_fixed_100000", "two_set_asym_80000_20000", "dam_break_sym_100000", "lattice_mixed_double_10000",
                                  "edge_empty_and_tiny"])
def test_multi_device_context_matches_golden(name, oracle):
    """tnsx_options.n_devices: ONE context that cuts every run into slabs, runs one engine per device and gathers the lists into one
    host view (the mode the C++ drop-in uses for more than one GPU).  Here three engines share GPU 0: the whole path -- balanced
    cuts, ghosts in the upload, candidates-only ghosts, global ids, gathered offsets -- against the reference-generated fixtures."""
    import treensearch_amd as T
    case = CS.by_name(name)
    ns = T.TreeNSearch(devices=[0, 0, 0])
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
        P.assert_matches_golden(res, load_golden(case.name), 0, oracle, f"{name} on three engines (run {step})")
    st = ns.get_stats()
    assert st["n_devices_used"] == (3 if case.n_total() > 1000 else st["n_devices_used"]) and st["n_neighbors"] == sum(int(res[pr][0][-1]) for pr in case.active)
    # z-sort through the same context: a permutation that is Morton-sorted on the reported world box
    ns.prepare_zsort()
    for s, p in enumerate(case.points):
        order = ns.get_zsort_order(s)
        assert np.array_equal(np.sort(order), np.arange(len(p)))


def test_cpp_dropin_scenarios_on_two_engines(built_library, tmp_path):
    """the reference's test scenarios through the C++ shim with TNSX_DEVICES=0,0 (multi-device mode, two engines on GPU 0)"""
    exe = tmp_path / "shim_scenarios"
    lib_dir = os.path.dirname(built_library)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_scenarios.cpp"), "-o", str(exe),
                           "-L" + lib_dir, "-ltnsx", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=dict(os.environ, TNSX_DEVICES="0,0"))
    print(out.stdout[-3000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "FAILED" not in out.stdout and "ALL PASSED" in out.stdout


def test_zsort_resolution_follows_the_reference(oracle):
    """prepare_zsort before any run(): the reference sorts the POINTS on the cell grid refined to just under 2^21 steps per axis
    (_compute_zsort_order_notree, TreeNSearch.cpp:2678-2699); after a run() it orders whole cells (:2603-2660).  Both orders are
    checked on the grid the engine reports, and the fine one must really be finer than the cell grid."""
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    pts = D.uniform_cloud(50000, 17)
    r = D.radius_for_neighbors(50000, 30.0)
    cell = np.float32(1.5) * np.float32(r)
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(pts)
    ns.set_active_search(0, 0, True)
    ns.prepare_zsort()                                   # no run() yet: point-level order
    st = ns.get_stats()
    inv_fine = np.float32(st["zsort_cell_size_inv"])
    world = np.float32(st["world_top"][0]) - np.float32(st["world_bottom"][0])
    steps = float(world) * float(inv_fine)
    assert 0.999 * 2.0 ** 20 <= steps < 2.0 ** 21 and float(inv_fine) > 1000.0 / float(cell)
    order = ns.get_zsort_order(0)
    assert np.array_equal(np.sort(order), np.arange(len(pts)))
    keys = oracle.zsort_keys(pts, np.array(st["world_bottom"], np.float32), inv_fine)
    assert oracle.check_zsort(keys, order) == 0
    assert len(np.unique(keys)) > 0.99 * len(pts)        # (at this resolution nearly every point has a key of its own)
    ns.run()
    ns.prepare_zsort()                                   # after a run(): cell-level order, in-cell order kept
    st = ns.get_stats()
    assert np.float32(st["zsort_cell_size_inv"]) == np.float32(1.0) / cell
    order = ns.get_zsort_order(0)
    keys = oracle.zsort_keys(pts, np.array(st["world_bottom"], np.float32), np.float32(1.0) / cell)
    assert oracle.check_zsort(keys, order) == 0
    same = keys[order][1:] == keys[order][:-1]
    assert np.all(np.diff(order)[same] > 0), "points of one cell must keep their order"
    ns.prepare_zsort()                                   # twice in a row: the cells are gone again (TreeNSearch.cpp:2659-2660)
    assert np.float32(ns.get_stats()["zsort_cell_size_inv"]) == inv_fine


@pytest.mark.parametrize("cells_per_axis,n", [(128, 220_000), (256, 220_000), (512, 220_000), (512, 3_000_000), (1024, 1_000_000)],
                         ids=["128", "256", "512", "512-3m", "1024-1m"])
def test_zsort_after_a_run_on_large_sets(cells_per_axis, n, oracle):
    """From 65 536 points on the cell-level order after a run() comes from the ranked pass on the high digit plus one workgroup per bucket (k_morton_place) for keys up to
    24 bits, and -- round 5 -- from {key, index} pairs through single-pass digit sorts (decoupled look-back, tnsx_build.hip "Z-order of WIDE keys") for 25 - 30 bits (a
    reference grid of 512 or 1024 cells per axis; 3 M points are 367 tiles whose totals travel through the look-back).  A cloud with a dense blob (buckets beyond what a
    workgroup keeps in registers) on each grid size: a permutation, Morton-monotone on the reference's grid, stable where the path promises it, and apply_zsort moves the
    points accordingly."""
    import treensearch_amd as T
    rng = np.random.default_rng(cells_per_axis)
    pts = rng.random((n, 3), dtype=np.float32)
    pts[:70_000] = np.float32(0.5) + (rng.random((70_000, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(0.05)
    pts[0] = 0.0; pts[1] = 1.0                                  # (the box is the unit cube whatever the draw)
    r = np.float32(1.0 / (1.5 * 0.78 * cells_per_axis))         # cell edge 1.5 r: 0.78 * cells_per_axis cells across the cube
    ns = T.TreeNSearch(); ns.set_search_radius(r)
    ns.add_point_set(pts); ns.set_active_search(0, 0, True)
    ns.run()
    ns.prepare_zsort()
    st = ns.get_stats()
    inv = np.float32(st["zsort_cell_size_inv"])
    steps = (np.float32(st["world_top"][0]) - np.float32(st["world_bottom"][0])) * inv
    assert cells_per_axis / 2 < float(steps) <= cells_per_axis, (float(steps), cells_per_axis)
    order = ns.get_zsort_order(0)
    assert np.array_equal(np.sort(order), np.arange(n)), "the order must be a permutation"
    keys = oracle.zsort_keys(pts, np.array(st["world_bottom"], np.float32), inv)
    assert oracle.check_zsort(keys, order) == 0
    if cells_per_axis >= 512:
        # the wide-key path is a STABLE sort of the points by the Morton code of their cell: points of one cell keep their input order
        ks = keys[order]
        same = ks[1:] == ks[:-1]
        assert bool(np.all(order[1:][same] > order[:-1][same])), "the single-pass digit sorts must be stable"
        assert int(same.sum()) > n // 100, "the cloud has cells with several points (else the check above checks nothing)"
    import torch
    d_p = torch.from_numpy(pts.copy()).cuda()
    ns.apply_zsort(0, d_p, 3)
    assert np.array_equal(d_p.cpu().numpy(), pts[order])


# ---------------------------------------------------------------------------------------------------------------------
# World box and z-sort grid against the REFERENCE's private world box (the `world` block of every fixture: domain_float read through
# oracle/ref_wrap.cpp after run() / run_scalar() / prepare_zsort() on a fresh instance; tests/golden/make_golden.py::make_world).
# ---------------------------------------------------------------------------------------------------------------------
def _fixture_box(w):
    return np.array([float.fromhex(v) for v in w["bottom"] + w["top"]], np.float32)


_WORLD_CASES = [c for c in CS.small_cases() if c.n_total() > 0]


@pytest.mark.parametrize("case", _WORLD_CASES, ids=[c.name for c in _WORLD_CASES])
def test_world_box_and_zsort_grid_equal_the_references(case, oracle):
    from conftest import load_golden
    world = load_golden(case.name)["world"]

    def engine_box(ns):
        st = ns.get_stats()
        return np.array(st["world_bottom"] + st["world_top"], np.float32), st

    # -- prepare_zsort() on a fresh engine: the no-tree path (TreeNSearch.cpp:2671-2699) -> box of the SIMD bounds, refined grid
    ns = P.make_engine(case, 0)
    ns.prepare_zsort()
    box, st = engine_box(ns)
    w = world["zsort"]
    assert np.array_equal(box, _fixture_box(w)), f"prepare_zsort: {box} != {_fixture_box(w)}"
    assert st["world_cells_pow2"] == w["cells_pow2"]
    cell = np.float32(float.fromhex(w["cell_size"]))
    fine = cell
    while np.float32(box[3] - box[0]) / np.float32(fine / np.float32(2.0)) < np.float32(2097151):
        fine = np.float32(fine / np.float32(2.0))
    assert np.float32(st["zsort_cell_size_inv"]) == np.float32(1.0) / fine
    for s, p in enumerate(case.points):
        if len(p):
            keys = oracle.zsort_keys(np.asarray(p, np.float32), _fixture_box(w)[:3], np.float32(1.0) / fine)   # keys on the FIXTURE's box
            assert oracle.check_zsort(keys, ns.get_zsort_order(s)) == 0
    if not case.tns_ok:
        return
    # -- run() on a fresh engine: _update_world_AABB_simd (:523-645), then the tree path of prepare_zsort (:2603-2660) on the cell grid
    ns = P.make_engine(case, 0)
    ns.run()
    box, st = engine_box(ns)
    w = world["run"]
    assert np.array_equal(box, _fixture_box(w)), f"run: {box} != {_fixture_box(w)}"
    assert st["world_cells_pow2"] == w["cells_pow2"]
    ns.prepare_zsort()
    box2, st = engine_box(ns)
    assert np.array_equal(box2, box) and np.float32(st["zsort_cell_size_inv"]) == np.float32(1.0) / cell
    for s, p in enumerate(case.points):
        if len(p):
            keys = oracle.zsort_keys(np.asarray(p, np.float32), _fixture_box(w)[:3], np.float32(1.0) / cell)
            assert oracle.check_zsort(keys, ns.get_zsort_order(s)) == 0
    # -- run_scalar() on a fresh engine: _update_world_AABB (:415-522), no origin
    ns = P.make_engine(case, 0)
    ns.run_scalar()
    box, st = engine_box(ns)
    w = world["run_scalar"]
    assert np.array_equal(box, _fixture_box(w)), f"run_scalar: {box} != {_fixture_box(w)}"
    assert st["world_cells_pow2"] == w["cells_pow2"]
    # a later run() on the same engine re-snaps only if the origin is outside the scalar box (the persistence rule, :474-482)
    ns.run()
    box3, _ = engine_box(ns)
    sb = _fixture_box(w)
    if np.all(sb[:3] <= 0) and np.all(sb[3:] >= 0):
        assert np.array_equal(box3, sb)
    else:
        assert np.array_equal(box3, _fixture_box(world["run"]))


def test_cloud_far_from_the_origin_hits_the_cell_limit_like_the_reference():
    """TreeNSearch.cpp:633-638 with the SIMD bounds of :564-590: run() snaps a box that contains the ORIGIN, so a small cloud far away
    from it needs more than 2^15 cells per axis and the reference exits; the engine reports TNSX_ERR_GRID_TOO_LARGE (5) and stays
    usable.  run_scalar() (tight box) accepts the same cloud, as the reference's does."""
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    pts = (D.uniform_cloud(5000, 5) + np.float32(2000.0)).astype(np.float32)
    r = np.float32(0.03)          # cell 0.045: 2000 / 0.045 = 44 k cells from the origin
    ns = T.TreeNSearch()
    ns.set_search_radius(r)
    ns.add_point_set(pts)
    ns.set_active_search(0, 0, True)
    with pytest.raises(T.TnsxError) as e:
        ns.run()
    assert e.value.status == 5 and "32768" in e.value.message
    ns.run_scalar()
    assert ns.get_stats()["world_cells_pow2"] <= 64


# ----------------------------------------------------------------------------------------------------------------------
# "data_ptr() in, CSR tensors out" -- device-side consumers of the lists
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["uniform_fixed_100000", "dam_break_sym_1000000", "two_set_asym_80000_20000", "edge_empty_and_tiny"])
@pytest.mark.parametrize("exact_layout", [False, True])
def test_torch_csr_out_equals_host_csr_and_golden(name, exact_layout, oracle):
    """neighbor_csr_torch (gap-free CSR built on the device, tnsx_pair_csr_device) and neighbor_records_torch (zero-copy views of the engine's offsets /
    records) against neighbor_csr (host) and the reference's fixture -- C1 and the 1 M-point dam break of configs[3], plus two sets and empty sets."""
    import torch
    case = CS.by_name(name)
    ns = P.make_engine(case, 0, device_inputs=True, exact_layout=exact_layout)
    ns.run()
    res = {}
    for pr in case.active:
        h_offs, h_idx = ns.neighbor_csr(*pr)                       # host path, lists ascending
        d_offs, d_idx = ns.neighbor_csr_torch(*pr, sort_each=True)
        assert d_offs.is_cuda and d_idx.is_cuda and d_offs.dtype == torch.int64 and d_idx.dtype == torch.int32
        assert np.array_equal(d_offs.cpu().numpy(), h_offs) and np.array_equal(d_idx.cpu().numpy(), h_idx), f"{name} {pr}: device CSR != host CSR"
        res[pr] = (d_offs.cpu().numpy(), d_idx.cpu().numpy())
        # the zero-copy record views: records[offsets[p]] = count, then the indices -- the same lists again, read on the device
        offs, recs = ns.neighbor_records_torch(*pr)
        v = ns.pair_view(*pr)
        assert offs.numel() == v.n_points and recs.numel() == v.n_records
        if v.n_points:
            assert offs.data_ptr() == v.offsets_device and recs.data_ptr() == v.records_device, "views, not copies"
            counts = recs[offs].to(torch.int64)
            assert torch.equal(counts, d_offs[1:] - d_offs[:-1])
            src = torch.repeat_interleave(offs + 1 - d_offs[:-1], counts) + torch.arange(int(d_offs[-1].item()), device="cuda", dtype=torch.int64)
            unsorted_offs, unsorted_idx = ns.neighbor_csr_torch(*pr)
            assert torch.equal(recs[src], unsorted_idx) and torch.equal(unsorted_offs, d_offs), "tnsx_pair_csr_device copies the records' own order"
    P.assert_matches_golden(res, load_golden(case.name), 0, oracle, f"{name} torch CSR")
    assert ns.get_stats()["nan_fixups"] == 0


@pytest.mark.parametrize("mirror,exact", [(False, False), (True, False), (False, True), (True, True)], ids=["pool", "pool-mirror", "exact", "exact-mirror"])
def test_nan_points_inside_the_query_range_have_empty_lists(mirror, exact, oracle):
    """A NaN x is "no point" (include/tnsx.h): it enters no cell, finds nothing and is found by nobody.  Scattered THROUGH the query range of a set searched in
    itself (round-5 advice: no query kernel ever wrote such a point's offset, and the compaction of the host mirror followed it), every such point must
    read as an EMPTY list -- through the device views, the device CSR and the gap-free host mirror -- and the others as if the NaN points did not exist."""
    import torch
    import treensearch_amd as T
    from treensearch_amd import datagen as D
    n = 60000
    pts = D.uniform_cloud(n, 99)
    r = D.radius_for_neighbors(n, 30.0)
    rng = np.random.default_rng(11)
    absent = np.sort(rng.choice(n, 300, replace=False))
    real = np.setdiff1d(np.arange(n), absent)
    bad = pts.copy()
    bad[absent, 0] = np.nan
    bad[absent[::2], 1] = np.float32(1.0e6)          # (whatever else such a row holds counts for nothing)
    ro, ri = oracle.pair_search(np.ascontiguousarray(pts[real]), np.ascontiguousarray(pts[real]), radius=r, same_set=True)
    want_counts = np.zeros(n, np.int64)
    want_counts[real] = np.diff(ro)
    want_offs = np.zeros(n + 1, np.int64)
    want_offs[1:] = np.cumsum(want_counts)
    want_idx = real[ri].astype(np.int32)              # (ascending inside every list: real is increasing)
    ns = T.TreeNSearch(mirror_to_host=mirror, exact_layout=exact)
    ns.set_search_radius(r)
    src = bad if mirror else torch.from_numpy(bad).cuda()
    ns.add_point_set(src)
    ns.set_active_search(0, 0, True)
    for step in range(3):                             # the dry pass + sized pass, then two steady-state runs on the reused grid
        ns.run()
        # pool layout: the records of the pass do not add up to neighbours + queries, so the stray offsets are pointed at the pool's empty record;
        # exact layout: every such point gets a record of its own (one int: count 0) in the scan
        assert ns.get_stats()["nan_fixups"] == (0 if exact else 1)
        assert ns.get_stats()["n_neighbors"] == int(want_offs[-1])
        offs, idx = ns.neighbor_csr(0, 0)
        assert np.array_equal(offs, want_offs) and np.array_equal(idx, want_idx), f"run {step}"
        d_offs, d_idx = ns.neighbor_csr_torch(0, 0, sort_each=True)
        assert np.array_equal(d_offs.cpu().numpy(), want_offs) and np.array_equal(d_idx.cpu().numpy(), want_idx)
        if mirror:
            import ctypes as C
            v = ns.pair_view(0, 0)
            ho = np.ctypeslib.as_array(C.cast(v.offsets_host, C.POINTER(C.c_uint64)), shape=(n,))
            hr = np.ctypeslib.as_array(C.cast(v.records_host, C.POINTER(C.c_int32)), shape=(int(want_offs[-1]) + n,))
            assert np.array_equal(hr[ho], want_counts) and int(ho[-1]) + 1 + int(want_counts[-1]) == int(want_offs[-1]) + n
            assert all(ns.get_neighborlist(0, 0, int(p)).size() == 0 for p in absent[:20])
