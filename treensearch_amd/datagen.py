"""Deterministic synthetic point clouds for tests and bench.py.

Counter-based (splitmix64) so that the same global point set is produced on any machine, for any
sharding of the index range, without depending on libstdc++ distribution semantics.  The workloads
follow SURVEY.md section 8(d) / BASELINE.json `configs`:

  C1/C2/C5  uniform_cloud(n, seed)                     U[0,1)^3, fixed radius for ~60 neighbours
  C3        two_set_cloud(n_fluid, n_boundary, seed)   fluid box + 2-layer lattice shell
  C4        dam_break_cloud(n, seed)                   clustered SPH-like cloud + per-point radii
  tests     sph_lattice(bottom, top, spacing)          the lattice of the reference's own tests
                                                       (/root/reference/tests/tests.cpp:16-32)
"""
from __future__ import annotations

import math

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output per 64-bit counter value (vectorised, wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        x = x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, start: int, count: int, stream: int = 0) -> np.ndarray:
    """float32 in [0,1): 24 random mantissa bits of splitmix64(seed, stream, start+i)."""
    with np.errstate(over="ignore"):
        base = np.uint64((seed * 0x100000001B3 + stream * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF)
        ctr = np.arange(start, start + count, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    z = splitmix64(ctr)
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def radius_for_neighbors(n: int, k: float = 60.0, volume: float = 1.0) -> np.float32:
    """r such that a ball of radius r holds k of n points spread uniformly over `volume`."""
    return np.float32((3.0 * k * volume / (4.0 * math.pi * n)) ** (1.0 / 3.0))


def uniform_cloud(n: int, seed: int = 12345, start: int = 0) -> np.ndarray:
    """Points start..start+n-1 of the global U[0,1)^3 cloud with this seed, shape (n,3) float32."""
    u = uniform01(seed, 3 * start, 3 * n)
    return np.ascontiguousarray(u.reshape(n, 3))


def _i64(c: int) -> int:
    """a 64-bit constant as the signed value with the same bits (torch has no uint64 arithmetic)"""
    c &= 0xFFFFFFFFFFFFFFFF
    return c - (1 << 64) if c >= (1 << 63) else c


def uniform_cloud_torch(n: int, seed: int = 12345, start: int = 0, device="cuda", chunk: int = 1 << 26):
    """uniform_cloud(n, seed, start) generated ON THE DEVICE, bit for bit the same points (tests/test_oracle_golden.py compares the two): splitmix64 in
    wrapping int64 arithmetic, logical shifts spelled as shift + mask.  200 M points take a second on the GPU instead of minutes of numpy on one host core --
    what lets the N = 1 bench line carry a live leg of configs[4] at its full size."""
    import torch
    out = torch.empty(3 * n, dtype=torch.float32, device=device)
    base = _i64(seed * 0x100000001B3)
    mul, gamma, m1, m2 = _i64(0xD1342543DE82EF95), _i64(0x9E3779B97F4A7C15), _i64(0xBF58476D1CE4E5B9), _i64(0x94D049BB133111EB)

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)
    for lo in range(0, 3 * n, chunk):
        hi = min(lo + chunk, 3 * n)
        z = torch.arange(3 * start + lo, 3 * start + hi, dtype=torch.int64, device=device) * mul + base + gamma
        z = (z ^ lsr(z, 30)) * m1
        z = (z ^ lsr(z, 27)) * m2
        z = z ^ lsr(z, 31)
        out[lo:hi] = lsr(z, 40).to(torch.float32) * (2.0 ** -24)
    return out.view(n, 3)


def sph_lattice(bottom, top, spacing: float) -> np.ndarray:
    """Regular lattice with fp32 running sums, exactly as the reference's test generator builds it
    (tests/tests.cpp:16-32: `for (float x = bottom; x <= top; x += d)`)."""
    d = np.float32(spacing)

    def axis(b, t):
        out = []
        x = np.float32(b)
        t = np.float32(t)
        while x <= t:
            out.append(x)
            x = np.float32(x + d)
        return np.array(out, dtype=np.float32)

    xs, ys, zs = axis(bottom[0], top[0]), axis(bottom[1], top[1]), axis(bottom[2], top[2])
    g = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), axis=-1).reshape(-1, 3)
    return np.ascontiguousarray(g.astype(np.float32))


def sph_lattice_for(n_points: int, scale: float = 1.0):
    """(points, search_radius) of the reference's `generate_point_grid_as_SPH({-1},{1}, d)` with
    d = scale * 2 / n^(1/3) (tests/tests.cpp:96-97, 119-121) and search radius 1.99 * d."""
    d = np.float32(np.float32(scale) * np.float32(2.0 / (float(n_points) ** (1.0 / 3.0))))
    pts = sph_lattice((-1, -1, -1), (1, 1, 1), d)
    return pts, np.float32(np.float32(1.99) * d)


def two_set_cloud(n_fluid: int, n_boundary: int, seed: int = 12345):
    """C3: fluid U in [0,1)x[0,0.8)x[0,1); boundary = 2-layer lattice shell just outside that box.

    Returns (fluid (n_fluid,3), boundary (n_boundary,3), radius).  The radius gives ~60 fluid-fluid
    neighbours."""
    f = uniform_cloud(n_fluid, seed)
    f[:, 1] *= np.float32(0.8)
    r = radius_for_neighbors(n_fluid, 60.0, 0.8)
    # shell: points on the 6 faces, two layers, spacing from the fluid mean spacing
    s = (0.8 / n_fluid) ** (1.0 / 3.0)
    lo = np.array([0.0, 0.0, 0.0]) - 2 * s
    hi = np.array([1.0, 0.8, 1.0]) + 2 * s
    ax = [np.arange(lo[d], hi[d] + 0.5 * s, s) for d in range(3)]
    g = np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3)
    inside = np.all((g > np.array([-0.5 * s] * 3)) & (g < np.array([1.0, 0.8, 1.0]) + 0.5 * s), axis=1)
    shell = g[~inside]
    if len(shell) >= n_boundary:
        sel = np.linspace(0, len(shell) - 1, n_boundary).astype(np.int64)
        b = shell[sel]
    else:  # top up with jittered copies so that the requested count is met
        reps = int(math.ceil(n_boundary / max(len(shell), 1)))
        b = np.tile(shell, (reps, 1))[:n_boundary]
        jit = (uniform01(seed + 1, 0, 3 * n_boundary).reshape(-1, 3) - 0.5) * (0.25 * s)
        b = b + jit
    return f, np.ascontiguousarray(b.astype(np.float32)), r


def dam_break_cloud(n: int, seed: int = 12345, neighbors: float = 40.0):
    """C4: 70 % jittered-lattice dense column (x<0.35, y<0.6), 25 % thin floor layer (y<0.08),
    5 % sparse spray over the unit cube; radii r_i = r0 * (1 + u_i), u ~ U[0,1) so r_max/r_min < 2.

    Returns (points (n,3) float32, radii (n,) float32, r0)."""
    n_col = int(0.70 * n)
    n_floor = int(0.25 * n)
    n_spray = n - n_col - n_floor
    # dense column: lattice with jitter +-0.2 spacing
    vol = 0.35 * 0.6 * 1.0
    s = (vol / max(n_col, 1)) ** (1.0 / 3.0)
    nx, ny = max(int(0.35 / s), 1), max(int(0.6 / s), 1)
    nz = int(math.ceil(n_col / (nx * ny)))
    idx = np.arange(n_col, dtype=np.int64)
    ix, iy, iz = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    col = np.stack([ix * (0.35 / nx), iy * (0.6 / ny), iz * (1.0 / max(nz, 1))], axis=1)
    col += (uniform01(seed, 0, 3 * n_col).reshape(-1, 3).astype(np.float64) - 0.5) * (0.4 * s)
    # floor layer
    fl = uniform01(seed, 0, 3 * n_floor, stream=1).reshape(-1, 3).astype(np.float64)
    fl[:, 0] = 0.35 + fl[:, 0] * 0.65
    fl[:, 1] *= 0.08
    # spray
    sp = uniform01(seed, 0, 3 * n_spray, stream=2).reshape(-1, 3).astype(np.float64)
    pts = np.concatenate([col, fl, sp], axis=0)
    pts = np.clip(pts, 0.0, 1.0).astype(np.float32)
    # r0: ~`neighbors` neighbours in the dense column at the mean radius 1.5 r0
    dens = n_col / vol
    r0 = np.float32(((3.0 * neighbors / (4.0 * math.pi * dens)) ** (1.0 / 3.0)) / 1.5)
    u = uniform01(seed, 0, n, stream=3)
    radii = (r0 * (np.float32(1.0) + u)).astype(np.float32)
    return np.ascontiguousarray(pts), np.ascontiguousarray(radii), r0
