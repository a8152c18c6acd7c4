#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REAL reference (oracle/_ref, built from /root/reference).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py [--large]

For every case of tests/cases.py and both builds of the reference
  * default   reference flags (GCC contracts the distance to FMAs)  -> arithmetic mode "contracted"
  * strict    the same + -ffp-contract=off                           -> arithmetic mode "strict"
the script runs tns::TreeNSearch::run() (AVX2 path) and, where N^2 is affordable, tests/BruteforceNSearch,
asserts that the two agree as per-point sets (that is the reference's own test, BruteforceNSearch.cpp:117-178),
cross-checks the CPU restatement (oracle/tns_oracle.c) in the matching arithmetic mode, and stores per active
pair: total, order-independent digest (see tnso_digest_csr), min/max count and the leading full lists.

The fixtures are DATA (inputs are regenerated from seeds by treensearch_amd.datagen; outputs are numbers).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases as CS  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = {"strict": (True, O.STRICT), "contracted": (False, O.CONTRACTED)}


def run_reference(case: CS.Case, strict: bool, orc: O.Oracle):
    """-> {(i,j): (offsets, indices)} in ORIGINAL index space, lists ascending."""
    ref = O.RefTreeNSearch(strict=strict)
    variable = case.radii is not None
    if not variable:
        ref.set_search_radius(case.radius)
    pts = [np.ascontiguousarray(p.copy()) for p in case.points]
    rad = [np.ascontiguousarray(r.copy()) for r in case.radii] if variable else [None] * len(pts)
    for p, r in zip(pts, rad):
        ref.add_point_set(p, r)
    for (i, j) in case.active:
        ref.set_active_search(i, j, True)
    ref.set_symmetric_search(case.symmetric)
    # z-sort the copies first (the reference expects spatially coherent input; tests.cpp:254-256)
    ref.prepare_zsort()
    perms = []
    for s, (p, r) in enumerate(zip(pts, rad)):
        perm = ref.get_zsort_order(s).copy()
        perms.append(perm)
        if len(p):
            ref.apply_zsort(s, p, 3)
            if r is not None:
                ref.apply_zsort(s, r, 1)
    ref.run()
    out = {}
    for (i, j) in case.active:
        offs_p, idx_p = ref.neighbor_csr(i, j, sort_each=False)
        out[(i, j)] = orc.remap_csr(perms[i], perms[j], offs_p, idx_p)
    return out


def run_bruteforce(case: CS.Case, strict: bool):
    bf = O.RefBruteforce(strict=strict)
    variable = case.radii is not None
    for s, p in enumerate(case.points):
        p32 = np.ascontiguousarray(p, np.float32)   # (float) cast == TreeNSearch.cpp:277-296
        if variable:
            bf.add_point_set(p32, np.ascontiguousarray(case.radii[s], np.float32))
        else:
            bf.add_point_set(p32, float(case.radius))
    for (i, j) in case.active:
        bf.set_active_search(i, j, True)
    bf.set_symmetric_search(case.symmetric)
    bf.run()
    return {(i, j): bf.neighbor_csr(i, j) for (i, j) in case.active}


def run_oracle(case: CS.Case, mode: int, orc: O.Oracle, pair):
    i, j = pair
    variable = case.radii is not None
    xa = np.ascontiguousarray(case.points[i], np.float32)
    xb = np.ascontiguousarray(case.points[j], np.float32)
    if variable:
        return orc.pair_search(xa, xb, ra=np.ascontiguousarray(case.radii[i], np.float32),
                               rb=np.ascontiguousarray(case.radii[j], np.float32), symmetric=case.symmetric,
                               same_set=(i == j), mode=mode)
    return orc.pair_search(xa, xb, radius=case.radius, same_set=(i == j), mode=mode)


def make(case: CS.Case, orc: O.Oracle) -> dict:
    t0 = time.time()
    fx = {
        "name": case.name, "note": case.note, "size_class": case.size_class,
        "n_points": [int(len(p)) for p in case.points],
        "radius_f32_hex": None if case.radius is None else float(case.radius).hex(),
        "variable_radii": case.radii is not None, "symmetric": bool(case.symmetric),
        "active": [list(p) for p in case.active],
        "input_checksum": [int(np.ascontiguousarray(p).view(np.uint8).astype(np.uint64).sum()) for p in case.points],
        "pairs": {},
        "checked_against": [],
    }
    for mode_name, (strict, omode) in MODES.items():
        if case.tns_ok:
            ref = run_reference(case, strict, orc)
            checked = ["tns::TreeNSearch::run"]
            if case.bruteforce_ok:
                bf = run_bruteforce(case, strict)
                for pr in case.active:
                    assert np.array_equal(bf[pr][0], ref[pr][0]) and np.array_equal(bf[pr][1], ref[pr][1]), \
                        f"{case.name}: reference TNS != reference BruteforceNSearch for pair {pr} ({mode_name})"
                checked.append("BruteforceNSearch::run")
        else:
            # outside the octree's valid regime: the reference's all-pairs search is the only truth
            assert case.bruteforce_ok
            ref = run_bruteforce(case, strict)
            checked = ["BruteforceNSearch::run"]
        for pr in case.active:
            offs, idx = ref[pr]
            oo, oi = run_oracle(case, omode, orc, pr)
            assert np.array_equal(oo, offs) and np.array_equal(oi, idx), \
                f"{case.name}: oracle != reference for pair {pr} ({mode_name})"
            cnt = np.diff(offs)
            dsum, dxor = orc.digest(offs, idx, already_sorted=True)
            key = f"{pr[0]}->{pr[1]}"
            ent = fx["pairs"].setdefault(key, {})
            k = min(case.full_lists, len(cnt))
            ent[mode_name] = {
                "total": int(offs[-1]), "digest_sum": f"{dsum:016x}", "digest_xor": f"{dxor:016x}",
                "min_count": int(cnt.min()) if len(cnt) else 0, "max_count": int(cnt.max()) if len(cnt) else 0,
                "first_counts": [int(c) for c in cnt[:k]],
                "first_lists": [[int(v) for v in idx[offs[p]:offs[p + 1]]] for p in range(k)],
            }
        fx["checked_against"] = checked + ["oracle/tns_oracle.c"]
    fx["modes_differ"] = any(e["strict"]["digest_sum"] != e["contracted"]["digest_sum"] for e in fx["pairs"].values())
    fx["world"] = make_world(case, orc)
    fx["generated_in_s"] = round(time.time() - t0, 2)
    return fx


def _fresh_reference(case: CS.Case, strict: bool):
    ref = O.RefTreeNSearch(strict=strict)
    variable = case.radii is not None
    if not variable:
        ref.set_search_radius(case.radius)
    pts = [np.ascontiguousarray(p.copy()) for p in case.points]
    rad = [np.ascontiguousarray(r.copy()) for r in case.radii] if variable else [None] * len(pts)
    for p, r in zip(pts, rad):
        ref.add_point_set(p, r)
    for (i, j) in case.active:
        ref.set_active_search(i, j, True)
    ref.set_symmetric_search(case.symmetric)
    return ref, pts


def _hex(a):
    return [float(v).hex() for v in np.asarray(a, np.float32)]


def fine_zsort_cell(world_size: np.float32, cell: np.float32) -> np.float32:
    """quantisation step of the no-tree z-sort, TreeNSearch.cpp:2686-2690 (fp32 throughout)"""
    cell = np.float32(cell)
    while np.float32(world_size) / np.float32(cell / np.float32(2.0)) < np.float32(2097151):
        cell = np.float32(cell / np.float32(2.0))
    return cell


def make_world(case: CS.Case, orc: O.Oracle) -> dict:
    """The reference's PRIVATE world box (TreeNSearch.h:400, read through oracle/ref_wrap.cpp) after each of the three entry
    points that update it, on a fresh instance over the case's points as generated:
      run          run()           -> _update_world_AABB_simd (TreeNSearch.cpp:523-645)
      run_scalar   run_scalar()    -> _update_world_AABB      (:415-522)
      zsort        prepare_zsort() -> no-tree path, _update_world_AABB_simd (:2671-2674)
    Asserts on the way: both builds of the reference agree; the oracle's restatement reproduces every box bit for bit; the
    reference's z-sort order is Morton-monotone under the oracle's keys on the REFERENCE's box, on both z-sort paths."""
    fm = np.finfo(np.float32).max
    out = {}
    for path in ("run", "run_scalar", "zsort"):
        if path != "zsort" and not case.tns_ok:
            continue      # (outside the octree's valid regime only the box of prepare_zsort is taken: it never builds the tree)
        boxes = []
        for strict in (False, True):
            ref, pts = _fresh_reference(case, strict)
            getattr(ref, "prepare_zsort" if path == "zsort" else path)()
            box, cell = ref.get_world_box(), ref.get_cell_size()
            boxes.append(box)
            n_total = sum(len(p) for p in pts)
            # -- the oracle's restatement
            tight = np.array([fm, fm, fm, -fm, -fm, -fm], np.float32)
            for p in pts:
                orc.tight_bounds(np.asarray(p, np.float32), tight, simd=(path != "run_scalar"))
            obox = np.array([fm, fm, fm, -fm, -fm, -fm], np.float32)
            n_pow2 = 0
            if n_total > 0:
                rc, n_pow2 = orc.world_box_update(obox, tight, cell)
                assert rc == 1
            assert np.array_equal(obox, box), f"{case.name} {path}: oracle world box != reference"
            # -- the reference's z-sort order under the oracle's keys, on the reference's box
            if n_total > 0 and path == "zsort":
                inv = np.float32(1.0) / fine_zsort_cell(box[3] - box[0], cell)
                for s, p in enumerate(pts):
                    keys = orc.zsort_keys(np.asarray(p, np.float32), box[:3], inv)
                    assert orc.check_zsort(keys, ref.get_zsort_order(s)) == 0, f"{case.name}: no-tree z-sort of set {s} not Morton-ordered"
            if n_total > 0 and path == "run":
                ref.prepare_zsort()                    # tree path (:2595-2660): cells ordered by the 32-bit code of their coordinates
                assert np.array_equal(ref.get_world_box(), box)
                inv = np.float32(1.0) / cell
                for s, p in enumerate(pts):
                    keys = orc.zsort_keys(np.asarray(p, np.float32), box[:3], inv)
                    assert orc.check_zsort(keys, ref.get_zsort_order(s)) == 0, f"{case.name}: tree z-sort of set {s} not Morton-ordered"
        assert np.array_equal(boxes[0], boxes[1]), f"{case.name} {path}: the two reference builds disagree on the world box"
        out[path] = {"bottom": _hex(boxes[0][:3]), "top": _hex(boxes[0][3:]), "cells_pow2": int(n_pow2), "cell_size": float(cell).hex()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true", help="also (re)generate the medium/large digest fixtures")
    ap.add_argument("--only", default=None)
    ap.add_argument("--world-only", action="store_true", help="only (re)write the `world` block of the existing fixtures")
    args = ap.parse_args()
    assert O.have_ref(), "oracle/_ref missing: run `make -C oracle ref` (needs /root/reference)"
    orc = O.Oracle()
    todo = CS.small_cases() + (CS.large_cases() if args.large else [])
    for case in todo:
        if args.only and args.only not in case.name:
            continue
        if args.world_only:
            path = os.path.join(HERE, case.name + ".json")
            with open(path) as f:
                fx = json.load(f)
            t0 = time.time()
            fx["world"] = make_world(case, orc)
            with open(path, "w") as f:
                json.dump(fx, f, separators=(",", ":"))
            print(f"{case.name}: world {fx['world'].get('run', fx['world']['zsort'])['cells_pow2']} cells, {time.time() - t0:.1f} s", flush=True)
            continue
        if case.size_class != "small":
            case.full_lists = min(case.full_lists, 64)
        fx = make(case, orc)
        path = os.path.join(HERE, case.name + ".json")
        with open(path, "w") as f:
            json.dump(fx, f, separators=(",", ":"))
        tot = {k: v["strict"]["total"] for k, v in fx["pairs"].items()}
        print(f"{case.name}: {fx['n_points']} totals {tot} modes_differ={fx['modes_differ']} "
              f"{fx['generated_in_s']} s -> {os.path.getsize(path)} B", flush=True)


if __name__ == "__main__":
    main()
