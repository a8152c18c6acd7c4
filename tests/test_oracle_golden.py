"""CPU suite: the oracle (oracle/tns_oracle.c) against the committed golden fixtures that were generated from the
real reference (tests/golden/make_golden.py), plus unit tests of the restated pieces."""
import os

import numpy as np
import pytest

import cases as CS
import parity as P
from conftest import load_golden

SMALL = CS.small_cases()


@pytest.mark.parametrize("case", SMALL, ids=[c.name for c in SMALL])
@pytest.mark.parametrize("mode", [0, 1], ids=["strict", "contracted"])
def test_oracle_matches_golden(case, mode, oracle):
    golden = load_golden(case.name)
    assert golden["n_points"] == [len(p) for p in case.points], "datagen drifted from the fixture"
    cs = [int(np.ascontiguousarray(p).view(np.uint8).astype(np.uint64).sum()) for p in case.points]
    assert golden["input_checksum"] == cs, "datagen produces different bytes than when the fixture was made"
    res = P.run_oracle_case(case, mode, oracle)
    P.assert_matches_golden(res, golden, mode, oracle, case.name)


@pytest.mark.parametrize("name", ["lattice_fixed_10000", "lattice_two_sets_var_10000", "edge_duplicates", "edge_far_outlier",
                                  "edge_empty_and_tiny", "edge_boundary_distance"])
def test_oracle_grid_equals_all_pairs(name, oracle):
    """The oracle's grid candidate generation never changes the sets (all-pairs restates BruteforceNSearch::run)."""
    case = CS.by_name(name)
    a = P.run_oracle_case(case, 0, oracle, use_grid=True)
    b = P.run_oracle_case(case, 0, oracle, use_grid=False)
    for pr in case.active:
        P.assert_same_csr(a[pr], b[pr], f"{name} {pr}")


def test_arithmetic_modes_are_what_they_claim(oracle):
    """STRICT = every op rounded; CONTRACTED = fma(dz,dz, fma(dx,dx, dy*dy)).  Checked against exact rational
    arithmetic on a point pair where the two differ."""
    from fractions import Fraction
    rng = np.random.default_rng(3)
    found = 0
    for _ in range(20000):
        p = rng.random(3, dtype=np.float32)
        q = (p + (rng.random(3, dtype=np.float32) - np.float32(0.5)) * np.float32(0.02)).astype(np.float32)
        s = oracle.dist_sq(p, q, 0)
        c = oracle.dist_sq(p, q, 1)
        d = [np.float32(p[k] - q[k]) for k in range(3)]
        strict = np.float32(np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1])) + np.float32(d[2] * d[2]))
        assert s == strict
        # fma with exact intermediate product, one rounding
        def fma(a, b, cc):
            return np.float32(float(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(cc))))
        contracted = fma(d[2], d[2], fma(d[0], d[0], np.float32(d[1] * d[1])))
        assert c == contracted
        found += int(s != c)
    assert found > 0, "the test never exercised a pair where the modes differ"


def test_world_box_restatement(oracle):
    """tnso_world_box_update follows TreeNSearch.cpp:474-521: cubic, power-of-two cells, persistent."""
    fm = np.finfo(np.float32).max
    box = np.array([fm, fm, fm, -fm, -fm, -fm], np.float32)
    pts = CS.uniform_fixed(100000).points[0]
    tight = oracle.tight_bounds(pts)
    rc, n = oracle.world_box_update(box, tight, 1.5 * 0.052322388)
    assert rc == 1 and n == 16
    length = box[3:] - box[:3]
    assert np.allclose(length, np.float32(1.5 * 0.052322388) * 16, rtol=1e-6)
    assert np.all(box[:3] <= tight[:3]) and np.all(tight[3:] <= box[3:])
    before = box.copy()
    rc2, _ = oracle.world_box_update(box, tight, 1.5 * 0.052322388)   # contained -> untouched
    assert rc2 == 0 and np.array_equal(before, box)
    huge = np.array([0, 0, 0, 1e6, 1, 1], np.float32)
    rc3, n3 = oracle.world_box_update(box, huge, 1.0)
    assert rc3 == -1 and n3 > 32768


def _fixture_box(w):
    return np.array([float.fromhex(v) for v in w["bottom"] + w["top"]], np.float32)


def _oracle_world(oracle, case, cell, simd):
    fm = np.finfo(np.float32).max
    tight = np.array([fm, fm, fm, -fm, -fm, -fm], np.float32)
    for p in case.points:
        oracle.tight_bounds(np.asarray(p, np.float32), tight, simd=simd)
    box = np.array([fm, fm, fm, -fm, -fm, -fm], np.float32)
    n_pow2 = 0
    if case.n_total() > 0:
        _, n_pow2 = oracle.world_box_update(box, tight, cell)
    return box, n_pow2


def _medium_case(name):
    # (built on demand: CS.large_cases() would generate the 10 M cloud at collection time)
    return {"uniform_fixed_1000000": lambda: CS.uniform_fixed(1000000, size_class="medium"),
            "two_set_asym_800000_200000": lambda: CS.two_set_asymmetric(800000, 200000, size_class="medium"),
            "dam_break_sym_1000000": lambda: CS.dam_break(1000000, True, size_class="medium")}[name]()


@pytest.mark.parametrize("case", SMALL + ["uniform_fixed_1000000", "two_set_asym_800000_200000", "dam_break_sym_1000000"],
                         ids=lambda c: c if isinstance(c, str) else c.name)
def test_world_box_equals_the_references_private_box(case, oracle):
    """The `world` block of every fixture is the reference's PRIVATE domain_float (TreeNSearch.h:400), read through
    oracle/ref_wrap.cpp after run() / run_scalar() / prepare_zsort() on a fresh instance.  The oracle's restatement -- tight
    bounds (united with the origin on the SIMD paths, TreeNSearch.cpp:564-569 + :587-590), then the cubic power-of-two snapping
    of :474-521 -- must reproduce it bit for bit, including the number of cells per axis."""
    if isinstance(case, str):
        case = _medium_case(case)
    world = load_golden(case.name)["world"]
    assert set(world) == ({"run", "run_scalar", "zsort"} if case.tns_ok else {"zsort"})
    for path, w in world.items():
        cell = np.float32(float.fromhex(w["cell_size"]))
        box, n_pow2 = _oracle_world(oracle, case, cell, simd=(path != "run_scalar"))
        assert np.array_equal(box, _fixture_box(w)), f"{case.name} {path}: {box} != {_fixture_box(w)}"
        assert n_pow2 == w["cells_pow2"]
    if case.tns_ok and case.n_total() > 0:
        # the origin is what separates the two paths: wherever the cloud does not contain it, the boxes must differ
        fm = np.finfo(np.float32).max
        tight = np.array([fm, fm, fm, -fm, -fm, -fm], np.float32)
        for p in case.points:
            oracle.tight_bounds(np.asarray(p, np.float32), tight)
        if np.any(tight[:3] > 0) or np.any(tight[3:] < 0):
            assert world["run"] != world["run_scalar"]
        assert world["run"] == world["zsort"]


@pytest.mark.parametrize("name", ["uniform_fixed_100000", "dam_break_sym_100000", "two_set_asym_80000_20000", "edge_duplicates",
                                  "lattice_mixed_double_10000", "edge_far_outlier"])
def test_reference_zsort_order_is_monotone_under_the_oracle_keys(name, oracle):
    """Runs the REAL reference (oracle/_ref; skipped where it was never built): its get_zsort_order must be Morton-ordered under
    tnso_zsort_keys evaluated on the reference's own box -- the no-tree path on the refined grid (TreeNSearch.cpp:2678-2699) and
    the tree path on the cell grid (:2603-2660).  This is what pins tnso_zsort_keys / the engine's z-sort grid alignment."""
    from oracle import oracle as O
    if not O.have_ref():
        pytest.skip("oracle/_ref was not built in this tree")
    case = CS.by_name(name)
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    # no-tree path
    ref, pts = mg._fresh_reference(case, strict=False)
    ref.prepare_zsort()
    box, cell = ref.get_world_box(), ref.get_cell_size()
    assert np.array_equal(box, _fixture_box(load_golden(name)["world"]["zsort"]))
    inv = np.float32(1.0) / mg.fine_zsort_cell(box[3] - box[0], cell)
    for s, p in enumerate(pts):
        if len(p):
            keys = oracle.zsort_keys(np.asarray(p, np.float32), box[:3], inv)
            assert oracle.check_zsort(keys, ref.get_zsort_order(s)) == 0
            # and the restated ORDER: the oracle's own z-sort gives the same key sequence (ties may be permuted)
            mine = oracle.zsort_order(np.asarray(p, np.float32), box[:3], inv)
            assert np.array_equal(keys[mine], keys[ref.get_zsort_order(s)])
    # tree path
    ref, pts = mg._fresh_reference(case, strict=False)
    ref.run()
    ref.prepare_zsort()
    box = ref.get_world_box()
    assert np.array_equal(box, _fixture_box(load_golden(name)["world"]["run"]))
    inv = np.float32(1.0) / cell
    for s, p in enumerate(pts):
        if len(p):
            keys = oracle.zsort_keys(np.asarray(p, np.float32), box[:3], inv)
            order = ref.get_zsort_order(s)
            assert oracle.check_zsort(keys, order) == 0     # (inside a cell the reference keeps the order of ITS cell lists: unspecified)


def test_morton_bit_order(oracle):
    """libmorton convention used at TreeNSearch.cpp:2617/2693: x -> bit 0, y -> bit 1, z -> bit 2."""
    assert oracle.morton3(1, 0, 0) == 1 and oracle.morton3(0, 1, 0) == 2 and oracle.morton3(0, 0, 1) == 4
    assert oracle.morton3(2, 0, 0) == 8 and oracle.morton3(3, 3, 3) == 63
    assert oracle.morton3(0x7fff, 0x7fff, 0x7fff) == (1 << 45) - 1


def test_zsort_oracle_is_a_morton_ordered_permutation(oracle):
    pts = CS.uniform_fixed(100000).points[0]
    fm = np.finfo(np.float32).max
    box = np.array([fm, fm, fm, -fm, -fm, -fm], np.float32)
    cell = np.float32(1.5) * np.float32(0.052322388)
    oracle.world_box_update(box, oracle.tight_bounds(pts, simd=True), cell)
    inv = np.float32(1.0) / cell
    order = oracle.zsort_order(pts, box[:3], inv)
    keys = oracle.zsort_keys(pts, box[:3], inv)
    assert oracle.check_zsort(keys, order) == 0
    assert oracle.check_zsort(keys, np.arange(len(pts), dtype=np.int32)) == 2
    bad = order.copy(); bad[0] = bad[1]
    assert oracle.check_zsort(keys, bad) == 1


def test_digest_is_order_independent(oracle):
    case = CS.by_name("lattice_fixed_100")
    offs, idx = P.run_oracle_case(case, 0, oracle)[(0, 0)]
    d0 = oracle.digest(offs, idx, already_sorted=True)
    rng = np.random.default_rng(0)
    shuffled = idx.copy()
    for p in range(len(offs) - 1):
        rng.shuffle(shuffled[offs[p]:offs[p + 1]])
    assert oracle.digest(offs, shuffled, already_sorted=False) == d0
    broken = idx.copy(); broken[0] += 1
    assert oracle.digest(offs, broken, already_sorted=False) != d0
