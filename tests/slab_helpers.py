"""Test infrastructure of the slab tests: a TreeNSearch-shaped stand-in backed by the CPU oracle (the product has no CPU search
path, so the gloo tests of treensearch_amd/multi.py inject this as their per-rank search backend) and the single-process truth
the unions of the slabs are compared with."""
import os

import numpy as np


def _np(t):
    return None if t is None else (t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t))


class OracleEngine:
    """The subset of treensearch_amd.TreeNSearch that SlabSearch uses, on oracle/tns_oracle.c."""

    def __init__(self):
        from oracle import oracle as O
        self.orc = O.Oracle()
        self.sets, self.active, self.radius, self.symmetric, self.res = [], {}, None, True, {}
        self.query_count, self.ids = {}, {}

    def set_search_radius(self, r): self.radius = np.float32(r)
    def set_symmetric_search(self, on): self.symmetric = bool(on)
    def add_point_set(self, pts, radii=None): self.sets.append((pts, radii)); return len(self.sets) - 1
    def resize_point_set(self, s, pts, radii=None): self.sets[s] = (pts, radii)
    def set_active_search(self, i, j, on=True): self.active[(i, j)] = bool(on)
    def set_query_count(self, s, n): self.query_count[s] = int(n)
    def set_point_ids(self, s, ids): self.ids[s] = ids

    def run(self):
        self.res = {}
        for (i, j), on in self.active.items():
            if not on:
                continue
            a, ra = _np(self.sets[i][0]).reshape(-1, 3), _np(self.sets[i][1])
            b, rb = _np(self.sets[j][0]).reshape(-1, 3), _np(self.sets[j][1])
            if ra is not None:
                offs, idx = self.orc.pair_search(a, b, ra=ra, rb=rb, symmetric=self.symmetric, same_set=(i == j))
            else:
                offs, idx = self.orc.pair_search(a, b, radius=self.radius, same_set=(i == j))
            nq = self.query_count.get(i, -1)
            if nq >= 0:                               # candidates-only tail: no lists for the points behind nq
                offs = offs[:nq + 1]
                idx = idx[:offs[-1]]
            ids = self.ids.get(j)
            if ids is not None:                       # user ids instead of indices
                idx = _np(ids).astype(np.int64)[idx]
            self.res[(i, j)] = (offs, idx)

    def neighbor_csr(self, i, j): return self.res[(i, j)]


def reference_lists(orc, sets, i, j, radius, symmetric=True):
    """(offsets, indices ascending) of pair (i -> j) on the WHOLE cloud; sets = [(points, radii or None)]"""
    a, ra = sets[i]
    b, rb = sets[j]
    if ra is not None:
        return orc.pair_search(a, b, ra=ra, rb=rb, symmetric=symmetric, same_set=(i == j))
    return orc.pair_search(a, b, radius=np.float32(radius), same_set=(i == j))


# ----------------------------------------------------------------------------------------------------------------------
# all slabs of a decomposition inside ONE process (one thread per emulated rank, one GPU): tests/test_gpu_slabs.py
# ----------------------------------------------------------------------------------------------------------------------
class LocalTransport:
    """Moves the messages of SlabExchange between the emulated ranks of one process: a FIFO per (source, destination).  The
    receiver's buffer must have exactly the shape of the message -- that IS the agreement the wire protocol promises."""

    def __init__(self, world: int, timeout: float = 60.0):   # a step takes milliseconds; the 0.8 - 280 s steps of round 3 were a bug (junk radii behind NaN x), not a slow host.
                                                             # A stalled step now fails within a minute with the per-rank step times and engine statistics (run_slabs_in_threads)
        import queue
        import threading
        self.q = {(s, d): queue.Queue() for s in range(world) for d in range(world) if abs(s - d) == 1}
        self.timeout = timeout
        self.world = world
        self._flags = [False] * world
        self._barrier = threading.Barrier(world)

    def any_flag(self, rank, flag):
        """all-reduce(max) of one flag over the emulated ranks (what SlabSearch uses torch.distributed for otherwise)"""
        self._flags[rank] = bool(flag)
        self._barrier.wait(timeout=self.timeout)
        out = any(self._flags)
        self._barrier.wait(timeout=self.timeout)     # nobody overwrites its flag before everybody has read
        return out

    def round_trip(self, rank, peers, out_msgs, in_msgs):
        for p in peers:
            if p in out_msgs:
                self.q[(rank, p)].put(out_msgs[p].clone())
        for p in peers:
            if p in in_msgs:
                msg = self.q[(p, rank)].get(timeout=self.timeout)
                assert tuple(msg.shape) == tuple(in_msgs[p].shape), f"rank {rank} expected {tuple(in_msgs[p].shape)} from {p}, got {tuple(msg.shape)}"
                in_msgs[p].copy_(msg)


def run_slabs_in_threads(world, make_slab, step_fn, n_steps=2):
    """Runs `n_steps` steps of `world` SlabSearch objects concurrently (thread k = rank k).  make_slab(k, transport) -> slab;
    step_fn(k, slab, step) performs one step.  -> list of slabs.  An exception in any thread is re-raised here."""
    import threading
    tr = LocalTransport(world)
    slabs, errors = [None] * world, []

    import time
    times = [[] for _ in range(world)]     # per rank: (step, seconds, engine stats of the step) -- evidence when a step stalls

    def work(k):
        try:
            slabs[k] = make_slab(k, tr)
            for s in range(n_steps):
                t0 = time.perf_counter()
                try:
                    step_fn(k, slabs[k], s)
                finally:
                    try:
                        st = slabs[k].engine.get_stats()
                        brief = {key: st[key] for key in ("ms_total", "ms_fill", "ms_sort", "ms_bounds", "pool_retries", "cold_passes", "speculation_redos", "grid_trimmed", "n_neighbors", "grid_dims", "n_occupied_cells")}
                    except Exception:   # noqa: BLE001
                        brief = None
                    times[k].append((s, round(time.perf_counter() - t0, 2), brief))
                    if times[k][-1][1] > 30.0:
                        import warnings
                        warnings.warn(f"slab rank {k} step {s} took {times[k][-1][1]} s: {brief}")
        except BaseException as e:   # noqa: BLE001 - reported to the main thread
            errors.append((k, e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(world)]
    import faulthandler
    faulthandler.dump_traceback_later(40, repeat=True)      # a stalled step: where every thread is, every 40 s (stderr; visible with -s or on failure)
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        faulthandler.cancel_dump_traceback_later()
    if os.environ.get("TNSX_TEST_VERBOSE"):
        print("slab step times per rank:", times, flush=True)
    if errors:
        raise RuntimeError(f"rank {errors[0][0]} failed: {errors[0][1]!r}; all errors: {[(k, repr(e)) for k, e in errors]}; step times per rank: {times}") from errors[0][1]
    return slabs


def union_csr(n_global, per_rank):
    """per_rank: [(gids, offsets, indices)] -> (offsets, indices) of all n_global points in global order"""
    counts = np.zeros(n_global, np.int64)
    for gids, offs, _ in per_rank:
        counts[gids] = np.diff(offs)
    g_offs = np.zeros(n_global + 1, np.int64)
    np.cumsum(counts, out=g_offs[1:])
    out = np.empty(int(g_offs[-1]), np.int64)
    for gids, offs, idx in per_rank:
        cnt = np.diff(offs)
        dst = np.repeat(g_offs[gids] - offs[:-1], cnt) + np.arange(len(idx), dtype=np.int64)
        out[dst] = idx
    return g_offs, out
