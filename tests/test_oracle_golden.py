"""CPU suite: the oracle (oracle/tns_oracle.c) against the committed golden fixtures that were generated from the
real reference (tests/golden/make_golden.py), plus unit tests of the restated pieces."""
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
    oracle.world_box_update(box, oracle.tight_bounds(pts), cell)
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
