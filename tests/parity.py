"""Helpers shared by the oracle tests (CPU) and the GPU parity tests."""
from __future__ import annotations

import numpy as np

import cases as CS

MODE_NAMES = {0: "strict", 1: "contracted"}


def run_oracle_case(case: CS.Case, mode: int, orc, use_grid: bool = True) -> dict:
    """{(i,j): (offsets, indices)} from the CPU restatement, original index space, lists ascending."""
    out = {}
    variable = case.radii is not None
    for (i, j) in case.active:
        xa = np.ascontiguousarray(case.points[i], np.float32)     # (float) cast == TreeNSearch.cpp:277-296
        xb = np.ascontiguousarray(case.points[j], np.float32)
        if variable:
            out[(i, j)] = orc.pair_search(xa, xb, ra=np.ascontiguousarray(case.radii[i], np.float32),
                                          rb=np.ascontiguousarray(case.radii[j], np.float32),
                                          symmetric=case.symmetric, same_set=(i == j), mode=mode, use_grid=use_grid)
        else:
            out[(i, j)] = orc.pair_search(xa, xb, radius=case.radius, same_set=(i == j), mode=mode, use_grid=use_grid)
    return out


def make_engine(case: CS.Case, arith: int, device_inputs: bool = False, **kw):
    """Builds a treensearch_amd.TreeNSearch configured like the reference tests configure tns::TreeNSearch."""
    import treensearch_amd as T
    ns = T.TreeNSearch(arith=arith, **kw)
    variable = case.radii is not None
    if not variable:
        ns.set_search_radius(case.radius)
    holders = []
    for s, p in enumerate(case.points):
        r = case.radii[s] if variable else None
        if device_inputs:
            import torch
            p = torch.from_numpy(np.ascontiguousarray(p)).cuda()
            r = torch.from_numpy(np.ascontiguousarray(r)).cuda() if r is not None else None
        holders.append((p, r))
        ns.add_point_set(p, r)
    for (i, j) in case.active:
        ns.set_active_search(i, j, True)
    ns.set_symmetric_search(case.symmetric)
    ns._case_holders = holders
    return ns


def run_engine_case(case: CS.Case, arith: int, device_inputs: bool = False, **kw) -> dict:
    ns = make_engine(case, arith, device_inputs, **kw)
    ns.run()
    return {pr: ns.neighbor_csr(*pr) for pr in case.active}, ns


def assert_same_csr(a, b, what: str):
    (oa, ia), (ob, ib) = a, b
    assert len(oa) == len(ob), f"{what}: different number of points"
    if not np.array_equal(oa, ob):
        bad = np.nonzero(np.diff(oa) != np.diff(ob))[0]
        raise AssertionError(f"{what}: neighbour counts differ at {len(bad)} points, first {bad[:5]}: "
                             f"{np.diff(oa)[bad[:5]]} vs {np.diff(ob)[bad[:5]]}")
    if not np.array_equal(ia, ib):
        k = int(np.nonzero(ia != ib)[0][0])
        p = int(np.searchsorted(oa, k, side="right") - 1)
        raise AssertionError(f"{what}: lists differ at point {p}: {ia[oa[p]:oa[p + 1]]} vs {ib[ob[p]:ob[p + 1]]}")


def assert_matches_golden(result: dict, golden: dict, mode: int, orc, what: str, lists_sorted: bool = True):
    """result {(i,j): (offsets, indices)} vs a fixture of tests/golden/.  lists_sorted=False: the lists are in any order (the
    digest routine then sorts every list itself, in C -- much faster than a numpy lexsort of 6e8 entries)."""
    m = MODE_NAMES[mode]
    for (i, j), (offs, idx) in result.items():
        g = golden["pairs"][f"{i}->{j}"][m]
        assert int(offs[-1]) == g["total"], f"{what} {i}->{j} [{m}]: total {int(offs[-1])} != golden {g['total']}"
        dsum, dxor = orc.digest(offs, idx, already_sorted=lists_sorted)
        assert f"{dsum:016x}" == g["digest_sum"] and f"{dxor:016x}" == g["digest_xor"], \
            f"{what} {i}->{j} [{m}]: digest mismatch"
        cnt = np.diff(offs)
        k = len(g["first_counts"])
        assert [int(c) for c in cnt[:k]] == g["first_counts"], f"{what} {i}->{j} [{m}]: leading counts differ"
        for p in range(k):
            assert sorted(int(v) for v in idx[offs[p]:offs[p + 1]]) == g["first_lists"][p], \
                f"{what} {i}->{j} [{m}]: list of point {p} differs"
        if len(cnt):
            assert int(cnt.min()) == g["min_count"] and int(cnt.max()) == g["max_count"]
