"""Seeded test inputs shared by the golden generator, the oracle tests and the GPU parity tests.

Each case mirrors a scenario of the reference's own tests (/root/reference/tests/tests.cpp) or one of
BASELINE.json's configs (scaled where noted).  Inputs come from treensearch_amd.datagen only, so they
are bit-reproducible on the GPU box.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from treensearch_amd import datagen as D


@dataclass
class Case:
    name: str
    points: List[np.ndarray]                 # per set, (n,3) float32 or float64
    radii: Optional[List[np.ndarray]]        # per set (n,) or None => fixed radius mode
    radius: Optional[np.float32]             # fixed radius or None
    active: List[Tuple[int, int]]            # (set_i searches in set_j)
    symmetric: bool = True
    bruteforce_ok: bool = True               # small enough for the O(N^2) reference oracle
    tns_ok: bool = True                      # inside the regime where the reference's octree itself is valid (SURVEY.md 8c:
                                             # ceil(r_max / cell) <= 2); False: the truth is BruteforceNSearch alone
    full_lists: int = 0                      # number of leading full lists kept in the fixture
    note: str = ""
    size_class: str = "small"                # small: CPU tests; medium/large: GPU / golden digests only

    def n_total(self) -> int:
        return int(sum(len(p) for p in self.points))


def _const(n, v):
    return np.full(n, np.float32(v), dtype=np.float32)


# ---------------------------------------------------------------- reference test scenarios
def one_set_fixed_radius(n_points: int) -> Case:
    """tests/tests.cpp:91-112"""
    pts, r = D.sph_lattice_for(n_points)
    return Case(f"lattice_fixed_{n_points}", [pts], None, r, [(0, 0)], full_lists=64)


def two_sets_variable_radius(n_points: int, scale1: float = 1.31) -> Case:
    """tests/tests.cpp:114-145 (radii constant per set; pairs 0->0, 0->1, 1->0)"""
    p0, r0 = D.sph_lattice_for(n_points)
    p1, r1 = D.sph_lattice_for(n_points, scale1)
    return Case(f"lattice_two_sets_var_{n_points}", [p0, p1], [_const(len(p0), r0), _const(len(p1), r1)], None,
                [(0, 0), (0, 1), (1, 0)], full_lists=32)


def mixed_float_double(n_points: int) -> Case:
    """tests/tests.cpp:147-186 (set 1 handed over as double)"""
    c = two_sets_variable_radius(n_points, 1.33)
    c.name = f"lattice_mixed_double_{n_points}"
    c.points[1] = c.points[1].astype(np.float64)
    c.radii[1] = c.radii[1].astype(np.float64)
    return c


# ---------------------------------------------------------------- BASELINE.json configs
def uniform_fixed(n: int, seed: int = 12345, size_class="small") -> Case:
    """C1 (100 k) / C2 (10 M) and scaled instances: U[0,1)^3, one set, r for ~60 neighbours."""
    return Case(f"uniform_fixed_{n}", [D.uniform_cloud(n, seed)], None, D.radius_for_neighbors(n), [(0, 0)],
                bruteforce_ok=n <= 120000, full_lists=256, size_class=size_class,
                note="BASELINE.json configs[0]/[1] shape")


def two_set_asymmetric(n_fluid: int, n_boundary: int, size_class="small") -> Case:
    """C3: fluid + static boundary, searches 0->0 and 0->1 only."""
    f, b, r = D.two_set_cloud(n_fluid, n_boundary)
    return Case(f"two_set_asym_{n_fluid}_{n_boundary}", [f, b], None, r, [(0, 0), (0, 1)],
                bruteforce_ok=(n_fluid + n_boundary) <= 120000, full_lists=128, size_class=size_class,
                note="BASELINE.json configs[2] shape")


def dam_break(n: int, symmetric: bool = True, size_class="small") -> Case:
    """C4: clustered cloud, per-point radii r0*(1+u), symmetric search."""
    p, rad, _ = D.dam_break_cloud(n)
    return Case(f"dam_break_{'sym' if symmetric else 'asym'}_{n}", [p], [rad], None, [(0, 0)], symmetric=symmetric,
                bruteforce_ok=n <= 120000, full_lists=128, size_class=size_class,
                note="BASELINE.json configs[3] shape")


def variable_two_sets_random(n0: int, n1: int, ratio: float = 2.5, symmetric: bool = True) -> Case:
    """Random coordinates + truly per-point radii on two sets, all four searches active
    (not covered by the reference's own tests; pinned by BruteforceNSearch).  ratio = r_max / r_min; above 3 the reference's
    octree is outside its valid regime (uint16 pivot underflow) and BruteforceNSearch alone is the truth --
    this engine must be exact for any ratio."""
    p0 = D.uniform_cloud(n0, 777)
    p1 = D.uniform_cloud(n1, 778) * np.float32(0.9) + np.float32(0.05)
    # ~30 neighbours at the smallest radius for the ratio of the first fixtures; wider ratios start smaller so that the lists at
    # the largest radius stay a few hundred entries long
    rbase = D.radius_for_neighbors(n0 + n1, 30.0 * min(1.0, (2.5 / ratio) ** 3 * 4.0))
    r0 = (rbase * (1.0 + (ratio - 1.0) * D.uniform01(779, 0, n0))).astype(np.float32)
    r1 = (rbase * (1.0 + (ratio - 1.0) * D.uniform01(780, 0, n1))).astype(np.float32)
    name = f"random_var_{'sym' if symmetric else 'asym'}_{n0}_{n1}" + ("" if ratio == 2.5 else f"_ratio{ratio:g}")
    return Case(name, [p0, p1], [r0, r1], None, [(0, 0), (0, 1), (1, 0), (1, 1)], symmetric=symmetric, full_lists=64 if ratio <= 3.0 else 8,
                tns_ok=ratio <= 3.0)


# ---------------------------------------------------------------- edge cases
def edge_duplicates() -> Case:
    """Coincident distinct points are neighbours of each other (README.md:63); self is excluded."""
    base = D.uniform_cloud(500, 99)
    pts = np.concatenate([base, base[:100], base[:50]], axis=0)
    return Case("edge_duplicates", [np.ascontiguousarray(pts)], None, np.float32(0.12), [(0, 0)], full_lists=32)


def edge_empty_and_tiny() -> Case:
    """Empty set beside non-empty ones; single-point set (tests.cpp:369, :453)."""
    a = D.uniform_cloud(300, 5)
    e = np.zeros((0, 3), np.float32)
    one = np.array([[0.5, 0.5, 0.5]], np.float32)
    return Case("edge_empty_and_tiny", [a, e, one], None, np.float32(0.2),
                [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (2, 2), (1, 1)], full_lists=16)


def edge_boundary_distance() -> Case:
    """Pairs at exactly d == r (inclusive `<=`) and one ulp beyond."""
    r = np.float32(0.25)
    pts = np.array([[0, 0, 0], [0.25, 0, 0], [0, np.nextafter(np.float32(0.25), np.float32(1)), 0],
                    [0, 0, -0.25], [0.5, 0.5, 0.5], [0.5, 0.75, 0.5], [3.0, 3.0, 3.0]], np.float32)
    return Case("edge_boundary_distance", [pts], None, r, [(0, 0)], full_lists=7)


def edge_far_outlier() -> Case:
    """A dense cluster plus far outliers: the grid is large and sparse relative to r."""
    a = D.uniform_cloud(2000, 11) * np.float32(0.05)
    out = np.array([[40.0, -3.0, 7.0], [-25.0, 60.0, 1.0], [40.0, -3.0, 7.005]], np.float32)
    return Case("edge_far_outlier", [np.ascontiguousarray(np.concatenate([a, out]))], None, np.float32(0.01),
                [(0, 0)], full_lists=16)


def small_cases() -> List[Case]:
    """Everything the CPU suite and the GPU parity suite run in full (oracle finishes in seconds)."""
    return [
        one_set_fixed_radius(1), one_set_fixed_radius(100), one_set_fixed_radius(10000),
        two_sets_variable_radius(100), two_sets_variable_radius(10000),
        mixed_float_double(10000),
        uniform_fixed(100000),
        two_set_asymmetric(80000, 20000),
        dam_break(100000, True), dam_break(100000, False),
        variable_two_sets_random(30000, 10000, 2.5, True), variable_two_sets_random(30000, 10000, 2.5, False),
        variable_two_sets_random(12000, 6000, 5.0, True), variable_two_sets_random(12000, 6000, 10.0, False),
        edge_duplicates(), edge_empty_and_tiny(), edge_boundary_distance(), edge_far_outlier(),
    ]


# Digest-only fixtures (generated once from the real reference): scaled / full BASELINE configs.  Built one at a time, on demand
# (the 10 M-point cloud takes a while to generate: nobody who asks for another case should pay for it)
_LARGE = {
    "uniform_fixed_1000000": lambda: uniform_fixed(1000000, size_class="medium"),
    "uniform_fixed_2000000": lambda: uniform_fixed(2000000, size_class="medium"),          # the scaled instance of configs[4] (slab tests)
    "two_set_asym_800000_200000": lambda: two_set_asymmetric(800000, 200000, size_class="medium"),
    "dam_break_sym_1000000": lambda: dam_break(1000000, True, size_class="medium"),
    "uniform_fixed_10000000": lambda: uniform_fixed(10000000, size_class="large"),
}


def large_cases() -> List[Case]:
    return [make() for make in _LARGE.values()]


_SMALL_CACHE: List[Case] = []


def by_name(name: str) -> Case:
    # (the small cases first, built once: large_cases() generates the 10 M cloud)
    if not _SMALL_CACHE:
        _SMALL_CACHE.extend(small_cases())
    for c in _SMALL_CACHE:
        if c.name == name:
            return c
    if name in _LARGE:
        c = _LARGE[name]()
        assert c.name == name
        return c
    raise KeyError(name)
