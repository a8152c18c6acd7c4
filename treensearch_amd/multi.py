"""Spatial-slab sharding of one neighbour search across the GPUs of a node.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI; "gloo" on CPU for the tests).  Rank k owns
the points of slab k along x.

  SlabDecomposition   where the slabs are: global AABB by all-reduce(min/max), one x-histogram per rank at cell-plane
                      granularity (tnsx_x_histogram), all-reduce(sum), cuts at the quantiles; redistribute() moves every
                      point to its owner once (all-to-all).
  SlabExchange        the ONE exchange of a step: every rank sends the points within one halo width (>= the largest
                      search radius) of its left / right slab face to that neighbour (tnsx_halo_pack + grouped
                      isend/irecv).  Steady state is speculative: fixed-capacity messages, the counts stay on the device
                      and are checked after the search has run, so the exchange itself never waits for the host.
  SlabSearch          any number of point sets with fixed or per-point radii.  The received ghosts are appended to the
                      owned points of their set, marked as candidates only (tnsx_set_query_count: they are found, but
                      get no lists of their own) and carry their global ids (tnsx_set_point_ids), so the unchanged
                      single-GPU engine emits GLOBAL neighbour ids directly; no collective touches the data path.

The result equals the single-device result on the union of all slabs (tests/test_distributed_cpu.py on gloo with an
injected CPU engine, tests/test_gpu_slabs.py on the HIP engine).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def slab_halo_masks(x: torch.Tensor, lo: float, hi: float, halo: float, has_left: bool, has_right: bool):
    """Boolean masks of the owned points that the left / right neighbour needs as ghosts."""
    left = (x < (lo + halo)) if has_left else torch.zeros_like(x, dtype=torch.bool)
    right = (x >= (hi - halo)) if has_right else torch.zeros_like(x, dtype=torch.bool)
    return left, right


# ======================================================================================================================
# decomposition
# ======================================================================================================================
class SlabDecomposition:
    """Balanced 1-D slabs along x at cell-plane granularity.

    cuts: float32 array of world + 1 ascending positions, cuts[0] = -inf, cuts[world] = +inf; rank k owns cuts[k] <= x < cuts[k+1].
    Interior cuts lie on multiples of `plane_width` above the global minimum and are at least one plane apart, so that with
    plane_width >= halo only ADJACENT slabs ever exchange ghosts."""

    MAX_PLANES = 32768   # the reference's cells-per-axis limit (TreeNSearch.cpp:510-515)

    def __init__(self, group=None, engine=None):
        self.group = group
        self.rank, self.world = _world(group)
        self.engine = engine if hasattr(engine, "x_histogram") else None

    def _all_reduce(self, t: torch.Tensor, op):
        if self.world > 1:
            dist.all_reduce(t, op=op, group=self.group)
        return t

    def global_bounds(self, point_sets: Sequence[torch.Tensor]):
        """(lo[3], hi[3]) float32 numpy arrays of the tight AABB of all points of all ranks: two all-reduces of 3 floats."""
        dev = point_sets[0].device
        lo = torch.full((3,), float("inf"), dtype=torch.float32, device=dev)
        hi = torch.full((3,), float("-inf"), dtype=torch.float32, device=dev)
        for p in point_sets:
            if p.shape[0]:
                lo = torch.minimum(lo, p.amin(dim=0))
                hi = torch.maximum(hi, p.amax(dim=0))
        self._all_reduce(lo, dist.ReduceOp.MIN)
        self._all_reduce(hi, dist.ReduceOp.MAX)
        return lo.cpu().numpy(), hi.cpu().numpy()

    def x_histogram(self, point_sets: Sequence[torch.Tensor], x0: float, plane_width: float, n_planes: int) -> torch.Tensor:
        """int64[n_planes]: points of ALL ranks per x plane (plane b = [x0 + b w, x0 + (b+1) w), clamped at both ends)."""
        dev = point_sets[0].device
        inv = float(np.float32(1.0) / np.float32(plane_width))
        if self.engine is not None and dev.type == "cuda":
            h32 = torch.zeros(n_planes, dtype=torch.int32, device=dev)
            for p in point_sets:
                if p.shape[0]:
                    self.engine.x_histogram(p, x0, inv, h32)
            self.engine.synchronize()
            hist = h32.to(torch.int64)
        else:
            hist = torch.zeros(n_planes, dtype=torch.int64, device=dev)
            for p in point_sets:
                if p.shape[0]:
                    b = ((p[:, 0] - np.float32(x0)) * np.float32(inv)).to(torch.int64).clamp_(0, n_planes - 1)   # same fp32 ops as the kernel
                    hist += torch.bincount(b, minlength=n_planes)
        return self._all_reduce(hist, dist.ReduceOp.SUM)

    def balanced_cuts(self, point_sets: Sequence[torch.Tensor], plane_width: float, bounds=None, n_slabs: Optional[int] = None) -> np.ndarray:
        """n_slabs: number of slabs to cut into (default: one per rank)"""
        lo, hi = bounds if bounds is not None else self.global_bounds(point_sets)
        return self._cuts(point_sets, plane_width, lo, hi, self.world if n_slabs is None else int(n_slabs))

    def _cuts(self, point_sets, plane_width, lo, hi, world):
        x0, x1 = float(lo[0]), float(hi[0])
        if not (math.isfinite(x0) and math.isfinite(x1)):      # no points anywhere
            x0, x1 = 0.0, 0.0
        n_planes = int((x1 - x0) / float(plane_width)) + 1
        if n_planes > self.MAX_PLANES:
            raise ValueError(f"{n_planes} x planes of width {plane_width}: more than {self.MAX_PLANES}")
        if n_planes < world:
            raise ValueError(f"the cloud spans {n_planes} cell planes along x, fewer than the {world} slabs asked for")
        hist = self.x_histogram(point_sets, x0, plane_width, n_planes).cpu().numpy()
        cum = np.cumsum(hist)
        total = int(cum[-1]) if len(cum) else 0
        cuts = np.empty(world + 1, np.float32)
        cuts[0], cuts[-1] = -np.inf, np.inf
        prev = 0
        for k in range(1, world):
            # the plane boundary whose count of points to its left is closest to k/world of all; at least one plane per slab
            if total:
                target = total * k / world
                b = int(np.searchsorted(cum, target, side="left")) + 1             # boundary b has cum[b - 1] points to its left
                if b >= 2 and abs(cum[b - 2] - target) <= abs(cum[min(b, n_planes) - 1] - target):
                    b -= 1
            else:
                b = k
            b = min(max(b, prev + 1), n_planes - (world - k))
            cuts[k] = np.float32(np.float32(x0) + np.float32(b) * np.float32(plane_width))
            prev = b
        return cuts

    @staticmethod
    def owner_of(x: torch.Tensor, cuts: np.ndarray) -> torch.Tensor:
        """slab index of every x (cuts[k] <= x < cuts[k+1])"""
        inner = torch.as_tensor(np.asarray(cuts[1:-1], np.float32), device=x.device)
        return torch.bucketize(x.contiguous(), inner, right=True)

    def redistribute(self, pts: torch.Tensor, gids: torch.Tensor, radii: Optional[torch.Tensor], cuts: np.ndarray):
        """Moves every point to the rank that owns its slab (one all-to-all of counts, one of rows).  -> (pts, gids, radii)"""
        if self.world == 1:
            return pts, gids, radii
        dev = pts.device
        cols = 5 if radii is None else 6
        owner = self.owner_of(pts[:, 0], cuts)
        order = torch.sort(owner, stable=True).indices
        rows = torch.empty((pts.shape[0], cols), dtype=torch.float32, device=dev)
        rows[:, 0:3] = pts.index_select(0, order)
        if radii is not None:
            rows[:, 3] = radii.index_select(0, order)
        rows[:, cols - 2:cols] = gids.index_select(0, order).view(-1, 1).view(torch.float32).view(-1, 2)
        send_counts = torch.bincount(owner, minlength=self.world).to(torch.int64)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        s_list, r_list = send_counts.cpu().tolist(), recv_counts.cpu().tolist()
        out = torch.empty((sum(r_list), cols), dtype=torch.float32, device=dev)
        dist.all_to_all_single(out.view(-1), rows.view(-1), [c * cols for c in r_list], [c * cols for c in s_list], group=self.group)
        o_pts = out[:, 0:3].contiguous()
        o_rad = out[:, 3].contiguous() if radii is not None else None
        o_gid = out[:, cols - 2:cols].clone(memory_format=torch.contiguous_format).view(torch.int64).view(-1)
        return o_pts, o_gid, o_rad


# ======================================================================================================================
# the exchange of one step
# ======================================================================================================================
class SlabExchange:
    """Ghost-halo exchange of ONE point set between neighbouring slabs.  Works on CPU tensors with gloo and CUDA tensors with RCCL.

    Wire format per neighbour and step: ONE message of `capacity + 1` rows of W floats -- row 0 carries the row count (int32
    bits), rows 1..count the points [x, y, z, (r,) gid_lo, gid_hi].  `capacity` is what both sides derived from an earlier step's
    count (grow-only, +25 %); both directions of a link apply the same rule to the same numbers, so sender and receiver always
    agree on the message size.
      exact mode        (first step, CPU tensors, after an overflow): the counts are read on the host; a selection that does not
                        fit the capacity adds a second round with the exact size.
      speculative mode  (device tensors + engine packer + known capacities): nothing is read on the host.  The message is the
                        fixed-capacity prefix of the pack buffer, the received rows past the count are turned into NaN points
                        (the engine ignores points whose x is NaN) and `validate()` -- called after the search, which has
                        synchronised anyway -- tells whether a capacity was exceeded; the step is then repeated in exact mode."""

    def __init__(self, slab_lo: float, slab_hi: float, halo: float, group=None, packer=None, transport=None, rank=None, world=None):
        """transport: an object with round_trip(rank, peers, out_msgs, in_msgs) that moves the messages instead of
        torch.distributed (tests/test_gpu_slabs.py runs all slabs of a decomposition inside one process on one GPU); rank and
        world then say which slab this is."""
        self.rank, self.world = _world(group)
        if rank is not None:
            self.rank, self.world = int(rank), int(world)
        self.transport = transport
        self.group = group
        self.lo, self.hi, self.halo = float(slab_lo), float(slab_hi), float(halo)
        self.bytes_sent = 0
        self.packer = packer if hasattr(packer, "halo_pack") else None
        self._send_buf = [None, None]
        self._counts = None
        self._caps = {}          # peer -> (rows I may send, rows it may send) agreed for the one-round exchange
        self._pending = None     # speculative mode: (counts_out device tensor, {peer: count_in 0-dim tensor})
        self.rounds_last = 0
        self.speculative_last = False

    @staticmethod
    def _capacity(count: int) -> int:
        return count + count // 4 + 256

    def _peers(self):
        return [p for p in (self.rank - 1, self.rank + 1) if 0 <= p < self.world]

    # ---------------------------------------------------------------------------------------------- packing
    def _ensure_send_bufs(self, dev, cols, rows_needed):
        for side in (0, 1):
            b = self._send_buf[side]
            if rows_needed[side] and (b is None or b.shape[1] != cols + 1 or b.device != dev or b.shape[0] < rows_needed[side]):
                self._send_buf[side] = torch.empty((rows_needed[side], cols + 1), dtype=torch.float32, device=dev)

    def _pack_device(self, pts, gids, radii, has_l, has_r, cols):
        """Both sides in one kernel launch.  The rows land behind one header row of buffers that are reused from step to step
        (grown when the selection, or the capacity agreed with the peer, does not fit), so the message of the one-round
        exchange is a plain prefix of the buffer.  Each side has its own capacity (tnsx_halo_pack takes both)."""
        dev = pts.device
        if self._counts is None or self._counts.device != dev:
            self._counts = torch.zeros(2, dtype=torch.int32, device=dev)
        n = int(pts.shape[0])
        peers = (self.rank - 1, self.rank + 1)
        want = (has_l, has_r)
        need = [(max(n // 32, 1024, self._caps.get(peers[s], (0, 0))[0]) + 1) if want[s] else 0 for s in (0, 1)]
        while True:
            self._ensure_send_bufs(dev, cols, need)
            bl = self._send_buf[0][1:] if has_l else None
            br = self._send_buf[1][1:] if has_r else None
            cl, cr = self.packer.halo_pack(pts, gids, radii, self.lo + self.halo, self.hi - self.halo, bl, br, self._counts)
            grown = False
            for side, c in enumerate((cl, cr)):
                if want[side] and c + 1 > self._send_buf[side].shape[0]:      # this side's own capacity was exceeded: its rows are incomplete
                    need[side] = c + c // 8 + 1024
                    grown = True
            if not grown:
                return (self._send_buf[0][:cl + 1] if has_l else None), (self._send_buf[1][:cr + 1] if has_r else None)

    def _message(self, peer, msg, rows):
        """The first `rows` rows of the buffer behind `msg` when it is that large (device path: the pack buffer itself, rows past
        the count are unspecified), else a copy padded to `rows`."""
        side = 0 if peer < self.rank else 1
        buf = self._send_buf[side]
        if buf is not None and buf.data_ptr() == msg.data_ptr() and buf.shape[0] >= rows:
            return buf[:rows]
        if msg.shape[0] >= rows:
            return msg[:rows]
        out = torch.empty((rows, msg.shape[1]), dtype=msg.dtype, device=msg.device)
        out[:msg.shape[0]] = msg
        return out

    def _round_trip(self, peers, out_msgs, in_msgs):
        if self.transport is not None:
            self.bytes_sent += sum(m.numel() * 4 for m in out_msgs.values())
            self.transport.round_trip(self.rank, peers, out_msgs, in_msgs)
            return
        ops = []
        for p in peers:
            if p in out_msgs:
                ops.append(dist.P2POp(dist.isend, out_msgs[p], p, self.group))
                self.bytes_sent += out_msgs[p].numel() * 4
            if p in in_msgs:
                ops.append(dist.P2POp(dist.irecv, in_msgs[p], p, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    @staticmethod
    def _split(g, cols, with_radii):
        ghost_pts = g[:, 0:3].contiguous()
        ghost_r = g[:, 3].contiguous() if with_radii else None
        # (clone, not contiguous(): an EMPTY slice counts as contiguous and keeps its odd storage offset, which int64 cannot view)
        ghost_gid = g[:, cols - 1:cols + 1].clone(memory_format=torch.contiguous_format).view(torch.int64).view(-1)
        return ghost_pts, ghost_gid, ghost_r

    # ---------------------------------------------------------------------------------------------- the exchange
    def can_speculate(self, pts) -> bool:
        peers = self._peers()
        return (self.packer is not None and pts.is_cuda and len(peers) > 0 and hasattr(self.packer, "synchronize")
                and all(p in self._caps for p in peers))

    def exchange(self, pts: torch.Tensor, gids: torch.Tensor, radii: Optional[torch.Tensor] = None, speculative: bool = False):
        """pts (n,3) float32, gids (n,) int64 global ids, radii (n,) float32 or None.
        -> (ghost_pts (m,3), ghost_gids (m,), ghost_radii (m,) or None).  In speculative mode m is the agreed capacity and the
        rows past the real count are NaN points; call validate() once the search has run."""
        dev = pts.device
        has_l, has_r = self.rank > 0, self.rank < self.world - 1
        cols = 4 if radii is None else 5
        W = cols + 1
        self._pending = None
        self.speculative_last = False
        if speculative and self.can_speculate(pts):
            return self._exchange_speculative(pts, gids, radii, has_l, has_r, cols)

        def pack(mask):   # torch path (CPU tensors / no engine): header row + payload rows
            sel = torch.nonzero(mask, as_tuple=False).squeeze(1)
            out = torch.empty((sel.numel() + 1, W), dtype=torch.float32, device=dev)
            out[1:, 0:3] = pts.index_select(0, sel)
            if radii is not None:
                out[1:, 3] = radii.index_select(0, sel)
            out[1:, cols - 1:cols + 1] = gids.index_select(0, sel).view(-1, 1).view(torch.float32).view(-1, 2)
            return out

        send = {}   # peer -> (count + 1, W) tensor, row 0 = header (possibly a prefix of a larger buffer)
        if self.packer is not None and pts.is_cuda and (has_l or has_r):
            sl, sr = self._pack_device(pts, gids, radii, has_l, has_r, cols)
            if has_l:
                send[self.rank - 1] = sl
            if has_r:
                send[self.rank + 1] = sr
        else:
            if has_l:
                send[self.rank - 1] = pack(pts[:, 0] < (self.lo + self.halo))
            if has_r:
                send[self.rank + 1] = pack(pts[:, 0] >= (self.hi - self.halo))
        peers = sorted(send.keys())
        n_out = {p: int(send[p].shape[0]) - 1 for p in peers}
        for p in peers:
            send[p].view(torch.int32)[0, 0] = n_out[p]

        # ---- round 1: header + as many rows as the agreed capacity holds (capacity 0 before the first exchange)
        out1, in1 = {}, {}
        for p in peers:
            cs, cr = self._caps.get(p, (0, 0))
            out1[p] = self._message(p, send[p], cs + 1)
            in1[p] = torch.empty((cr + 1, W), dtype=torch.float32, device=dev)
        self._round_trip(peers, out1, in1)
        n_in = {p: int(in1[p].view(torch.int32)[0, 0].item()) for p in peers}
        # ---- round 2 (first step / overflow only): the full payload, now that both sides know the count
        out2 = {p: send[p][1:] for p in peers if n_out[p] > self._caps.get(p, (0, 0))[0]}
        in2 = {p: torch.empty((n_in[p], W), dtype=torch.float32, device=dev) for p in peers if n_in[p] > self._caps.get(p, (0, 0))[1]}
        if out2 or in2:
            self._round_trip(peers, out2, in2)
        recv = []
        for p in peers:
            recv.append(in2[p] if p in in2 else in1[p][1:1 + n_in[p]])
            self._grow_caps(p, n_out[p], n_in[p])
        self.rounds_last = 2 if (out2 or in2) else (1 if peers else 0)
        if not recv:
            return (torch.empty((0, 3), dtype=torch.float32, device=dev), torch.empty(0, dtype=torch.int64, device=dev),
                    None if radii is None else torch.empty(0, dtype=torch.float32, device=dev))
        g = torch.cat(recv, dim=0) if len(recv) > 1 else recv[0]
        return self._split(g, cols, radii is not None)

    def _grow_caps(self, p, n_out, n_in):
        cs, cr = self._caps.get(p, (0, 0))
        self._caps[p] = (max(cs, self._capacity(n_out)) if n_out > cs else cs, max(cr, self._capacity(n_in)) if n_in > cr else cr)

    def _exchange_speculative(self, pts, gids, radii, has_l, has_r, cols):
        dev = pts.device
        W = cols + 1
        if self._counts is None or self._counts.device != dev:
            self._counts = torch.zeros(2, dtype=torch.int32, device=dev)
        peers = self._peers()
        side_of = {self.rank - 1: 0, self.rank + 1: 1}
        self._ensure_send_bufs(dev, cols, [(self._caps[self.rank - 1][0] + 1) if has_l else 0, (self._caps[self.rank + 1][0] + 1) if has_r else 0])
        cap_s = {p: self._caps[p][0] for p in peers}
        cap_r = {p: self._caps[p][1] for p in peers}
        bl = self._send_buf[0][1:1 + cap_s[self.rank - 1]] if has_l else None
        br = self._send_buf[1][1:1 + cap_s[self.rank + 1]] if has_r else None
        self.packer.halo_pack(pts, gids, radii, self.lo + self.halo, self.hi - self.halo, bl, br, self._counts, wait=False)
        self.packer.order_after_engine()     # torch's stream continues behind the pack kernel (no host wait)
        out1, in1 = {}, {}
        for p in peers:
            s = side_of[p]
            self._send_buf[s].view(torch.int32)[0, 0:1].copy_(self._counts[s:s + 1])       # header = count, device to device
            out1[p] = self._send_buf[s][:cap_s[p] + 1]
            in1[p] = torch.empty((cap_r[p] + 1, W), dtype=torch.float32, device=dev)
        self._round_trip(peers, out1, in1)
        recv, cnt_in = [], {}
        for p in peers:
            c = in1[p].view(torch.int32)[0, 0]
            rows = in1[p][1:]
            absent = torch.arange(cap_r[p], device=dev, dtype=torch.int32) >= c
            rows[:, 0].masked_fill_(absent, float("nan"))                                   # x = NaN: no point (ignored by the engine)
            recv.append(rows)
            cnt_in[p] = c
        self._pending = (self._counts.clone(), cnt_in, cap_s, cap_r)
        self.rounds_last = 1
        self.speculative_last = True
        g = torch.cat(recv, dim=0) if len(recv) > 1 else recv[0]
        return self._split(g, cols, radii is not None)

    def validate(self) -> bool:
        """After a speculative exchange (and after the search has run): False if a capacity was exceeded -- the capacities have
        then been raised and the step must be repeated (in exact mode)."""
        if self._pending is None:
            return True
        counts_out, cnt_in, cap_s, cap_r = self._pending
        self._pending = None
        peers = sorted(cnt_in.keys())
        vals = torch.cat([counts_out.view(-1)] + [cnt_in[p].view(1) for p in peers]).tolist()
        ok = True
        for k, p in enumerate(peers):
            n_out = int(vals[0 if p < self.rank else 1])
            n_in = int(vals[2 + k])
            if n_out > cap_s[p] or n_in > cap_r[p]:
                ok = False
            self._grow_caps(p, n_out, n_in)
        return ok


# ======================================================================================================================
# the search of one slab
# ======================================================================================================================
class _SlabSet:
    def __init__(self):
        self.ex: Optional[SlabExchange] = None
        self.set_id: Optional[int] = None
        self.buf = None          # (capacity, 3) float32: owned points first, ghosts behind them
        self.rbuf = None         # (capacity,)   float32 radii, same layout (per-point radii only)
        self.ids = None          # (capacity,)   int32 global ids, same layout
        self.n_owned = 0
        self.n_ghost = 0
        self.owned_gids = None
        self.ghost_gids = None


class SlabSearch:
    """Owned + ghost search of one slab over any number of point sets.  `engine_factory()` must return an object with the
    TreeNSearch API (treensearch_amd.TreeNSearch on a GPU; the CPU tests inject an oracle-backed stand-in).

    The ghosts of a set are APPENDED to its owned points -- one point set [owned | ghosts] per user set -- so a step costs one
    engine run over n + m points (m = a few per cent of n) instead of extra (owned -> ghost) pairs that would visit every owned
    cell again only to find no ghost near it.  The tail is marked candidates-only (`set_query_count`): ghosts are found but
    get no lists.  The engine is given the global ids of all points (`set_point_ids`), so the lists it writes already hold
    global ids: `neighbors_device()` hands out the device view as it is.

    radius: fixed search radius of all sets, or None for per-point radii; then `max_radius` must bound every radius of every
    rank (it sizes the halo; checked on the owned radii every step)."""

    def __init__(self, slab_lo: float, slab_hi: float, radius: Optional[float], engine_factory: Callable[[], object],
                 halo_margin: float = 1.0e-3, group=None, max_radius: Optional[float] = None, speculative: bool = True,
                 transport=None, rank=None, world=None):
        self.group = group
        self._ex_args = dict(transport=transport, rank=rank, world=world)
        self.lo, self.hi = float(slab_lo), float(slab_hi)
        self.radius = None if radius is None else float(radius)
        self.variable = radius is None
        if self.variable and max_radius is None:
            raise ValueError("per-point radii: max_radius (an upper bound of every search radius) is needed to size the halo")
        self.max_radius = float(max_radius) if self.variable else self.radius
        self.halo = self.max_radius * (1.0 + halo_margin)
        self.engine = engine_factory()
        if not self.variable:
            self.engine.set_search_radius(radius)
        self.speculative = bool(speculative)
        self.sets: List[_SlabSet] = []
        self.redone_last = False
        self._radius_check = None

    # -------- compatibility with the single-set use of round 1
    @property
    def ex(self) -> SlabExchange:
        return self._set(0).ex

    @property
    def ghost_gids(self):
        return self.sets[0].ghost_gids

    @property
    def n_owned(self):
        return self.sets[0].n_owned

    def _set(self, k: int) -> _SlabSet:
        while len(self.sets) <= k:
            s = _SlabSet()
            s.ex = SlabExchange(self.lo, self.hi, self.halo, self.group, packer=self.engine, **self._ex_args)
            self.sets.append(s)
        return self.sets[k]

    def set_symmetric_search(self, active: bool) -> None:
        self.engine.set_symmetric_search(active)

    def set_active_search(self, i: int, j: int, active: bool = True) -> None:
        self._set(max(i, j))
        self._active = getattr(self, "_active", {})
        self._active[(i, j)] = bool(active)
        if self.sets[i].set_id is not None and self.sets[j].set_id is not None:
            self.engine.set_active_search(self.sets[i].set_id, self.sets[j].set_id, bool(active))

    def owned_buffer(self, n: int, device, ghost_capacity: int = 0, set_index: int = 0) -> torch.Tensor:
        """(n,3) view of the internal point buffer of one set.  A caller that keeps its positions in this view saves step() the
        copy of the owned points (as long as the ghosts fit behind them; otherwise the buffer is re-allocated and step() copies)."""
        s = self._set(set_index)
        cap = n + max(int(ghost_capacity), n // 16, 1024)
        if s.buf is None or s.buf.shape[0] < cap or s.buf.device != torch.device(device):
            s.buf = torch.empty((cap, 3), dtype=torch.float32, device=device)
        return s.buf[:n]

    # -------- one step
    def step(self, *sets):
        """One exchange + one run.  Either step(pts, gids[, radii]) for a single set, or step((pts, gids[, radii]), ...) with one
        tuple per set: pts (n,3) float32 owned points, gids (n,) int64 global ids, radii (n,) float32 with per-point radii."""
        if sets and torch.is_tensor(sets[0]):
            sets = (tuple(sets),)
        sets = [tuple(s) + (None,) * (3 - len(s)) for s in sets]
        self.redone_last = False
        spec = self.speculative and all(self._set(k).ex.can_speculate(s[0]) for k, s in enumerate(sets))
        self._step_once(sets, spec)
        if spec:
            # validate() of EVERY set (it also raises the capacities it found too small: no short-circuit), then ONE agreement over
            # all ranks: the repeated step exchanges with both neighbours, so a rank whose own links were fine must repeat it too
            # -- left to each rank alone, an overflow on one link of a chain of >= 3 slabs would leave the neighbours' messages
            # unmatched (or matched with the NEXT step's message of the same shape).
            oks = [self.sets[k].ex.validate() for k in range(len(sets))]
            if self._agree_any(not all(oks), sets[0][0].device):
                # a capacity was exceeded somewhere: some ghosts are missing.  Once more, with the counts read on the host.
                self.redone_last = True
                self._step_once(sets, False)
        if self._radius_check is not None:
            bad = bool(self._radius_check.item())
            self._radius_check = None
            if bad:
                raise ValueError(f"a search radius exceeds max_radius = {self.max_radius}: the halo is too thin for exact results")

    def _agree_any(self, flag: bool, device) -> bool:
        """True on every rank if `flag` is set on any rank (one 4-byte all-reduce; an injected transport provides any_flag())."""
        tr = self._ex_args.get("transport")
        if tr is not None:
            return bool(tr.any_flag(self._set(0).ex.rank, bool(flag)))
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item())

    def _step_once(self, sets, speculative: bool):
        e = self.engine
        first = any(self._set(k).set_id is None for k in range(len(sets)))
        for k, (pts, gids, radii) in enumerate(sets):
            s = self._set(k)
            if self.variable != (radii is not None):
                raise ValueError("per-point radii must be given for every set, or for none (fixed radius)")
            ghost_pts, ghost_gid, ghost_r = s.ex.exchange(pts, gids, radii, speculative=speculative)
            n, m = int(pts.shape[0]), int(ghost_pts.shape[0])
            s.n_owned, s.n_ghost, s.owned_gids, s.ghost_gids = n, m, gids, ghost_gid
            dev = pts.device
            if s.buf is None or s.buf.shape[0] < n + m or s.buf.device != dev:
                s.buf = torch.empty((n + m + max((n + m) // 16, 1024), 3), dtype=torch.float32, device=dev)
            cap = s.buf.shape[0]
            if pts.data_ptr() != s.buf.data_ptr():
                s.buf[:n].copy_(pts)
            if m:
                s.buf[n:n + m].copy_(ghost_pts)
            if s.ids is None or s.ids.shape[0] < cap or s.ids.device != dev:
                s.ids = torch.empty(cap, dtype=torch.int32, device=dev)
            if getattr(s, "_ids_of", None) != (gids.data_ptr(), n, s.ids.data_ptr()):   # the owned ids rarely change: converted once
                s.ids[:n].copy_(gids)
                s._ids_of = (gids.data_ptr(), n, s.ids.data_ptr())
            if m:
                s.ids[n:n + m].copy_(ghost_gid)
            r_view = None
            if radii is not None:
                if s.rbuf is None or s.rbuf.shape[0] < cap or s.rbuf.device != dev:
                    s.rbuf = torch.empty(cap, dtype=torch.float32, device=dev)
                s.rbuf[:n].copy_(radii)
                if m:
                    s.rbuf[n:n + m].copy_(ghost_r)
                r_view = s.rbuf[:n + m]
                if n:
                    chk = radii.max() > self.max_radius
                    self._radius_check = chk if self._radius_check is None else (self._radius_check | chk)
            view = s.buf[:n + m]
            if s.set_id is None:
                s.set_id = e.add_point_set(view, r_view)
            else:
                e.resize_point_set(s.set_id, view, r_view)
            e.set_query_count(s.set_id, n)
            e.set_point_ids(s.set_id, s.ids[:n + m])
        if first:
            for (i, j), on in getattr(self, "_active", {(0, 0): True}).items():
                e.set_active_search(self.sets[i].set_id, self.sets[j].set_id, on)
        e.run()

    # -------- results
    def neighbors_device(self, i: int = 0, j: int = 0):
        """Device view (tnsx_csr_view) of the lists of the owned points of set i in set j: global ids, nothing left to translate."""
        return self.engine.pair_view(self.sets[i].set_id, self.sets[j].set_id)

    def global_neighbors(self, i: int = 0, j: int = 0):
        """(offsets int64[n+1], global ids int64[E]) of the owned points, every list ascending (host arrays; for the tests)."""
        offs, idx = self.engine.neighbor_csr(self.sets[i].set_id, self.sets[j].set_id)
        n = self.sets[i].n_owned
        offs = np.asarray(offs[:n + 1], dtype=np.int64)
        out = np.asarray(idx[offs[0]:offs[n]], dtype=np.int64)
        offs = offs - offs[0]
        lid = np.repeat(np.arange(n), np.diff(offs))
        order = np.lexsort((out, lid))
        return offs, out[order]


# ======================================================================================================================
# The slab layer BEHIND THE C ABI (include/tnsx.h "slab layer", treensearch_amd/csrc/tnsx_slab.cpp): the same step -- pack, one
# exchange, ghosts appended as candidates-only points with global ids, search, capacities checked afterwards -- done by
# libtnsx.so itself with RCCL (ncclSend / ncclRecv in one group per step); this class only hands over pointers.  A C++ consumer
# (SPlisHSPlasH + MPI) calls the same entry points.
# ======================================================================================================================
class SlabTransportC:
    """A tnsx_slab_transport.  rccl(): one communicator per rank, the 128-byte unique id travels over torch.distributed (any
    backend) or is given; local(group, rank): all slabs in one process (tests on one GPU)."""

    def __init__(self):
        from . import api as A
        self._A = A
        self._L = A.load_library()
        self.t = A.SlabTransport()
        self.kind = None

    @classmethod
    def rccl(cls, rank: int, world: int, device: int = -1, unique_id: Optional[bytes] = None, group=None):
        import ctypes as C
        self = cls()
        if unique_id is None:
            buf = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                raw = (C.c_ubyte * 128)()
                if self._L.tnsx_slab_rccl_unique_id(raw) != 0:
                    raise RuntimeError("tnsx_slab_rccl_unique_id failed: " + (self._L.tnsx_slab_rccl_error() or b"").decode())
                buf = torch.tensor(list(raw), dtype=torch.uint8)
            if world > 1:
                dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
                buf = buf.to(dev)
                dist.broadcast(buf, src=0, group=group)
                buf = buf.cpu()
            unique_id = bytes(buf.tolist())
        raw = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        st = self._L.tnsx_slab_transport_rccl(raw, int(rank), int(world), int(device), C.byref(self.t))
        if st != 0:
            raise RuntimeError("tnsx_slab_transport_rccl failed: " + (self._L.tnsx_slab_rccl_error() or b"").decode())
        self.kind = "rccl"
        return self

    @classmethod
    def local(cls, group_handle, rank: int):
        import ctypes as C
        self = cls()
        if self._L.tnsx_slab_transport_local(group_handle, int(rank), C.byref(self.t)) != 0:
            raise RuntimeError("tnsx_slab_transport_local failed")
        self.kind = "local"
        return self

    @classmethod
    def host_staged(cls, rank: int, world: int, group=None):
        """A transport over torch.distributed with the messages staged through host memory (any backend that moves CPU tensors: gloo):
        an application-filled tnsx_slab_transport, as include/tnsx.h invites MPI codes to write.  Slow (two PCIe crossings per message) but
        it needs neither RCCL nor peer access -- the ranks may even share one GPU -- and it is what tests/test_gpu_slabs.py runs the C
        entry points over real process boundaries with."""
        import ctypes as C
        from . import api as A
        self = cls()
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        H2D, D2H = 1, 2

        def exchange(user, rk, wd, ops, n_ops, stream):
            try:
                if hip.hipStreamSynchronize(stream) != 0:        # the send buffers are produced on the stream
                    return 1
                work, recvs = [], []
                for k in range(n_ops):
                    op = ops[k]
                    if op.send_bytes:
                        t = torch.empty(op.send_bytes, dtype=torch.uint8)
                        if hip.hipMemcpy(t.data_ptr(), op.send, op.send_bytes, D2H) != 0:
                            return 1
                        work.append(dist.isend(t, op.peer, group=group))
                    if op.recv_bytes:
                        t = torch.empty(op.recv_bytes, dtype=torch.uint8)
                        work.append(dist.irecv(t, op.peer, group=group))
                        recvs.append((op.recv, t))
                for w in work:
                    w.wait()
                for dst, t in recvs:                              # (synchronous copies: complete when the call returns, as the contract asks)
                    if hip.hipMemcpy(dst, t.data_ptr(), t.numel(), H2D) != 0:
                        return 1
                return 0
            except Exception:                                     # pragma: no cover - reported through the status
                import traceback
                traceback.print_exc()
                return 1

        def allreduce(user, rk, wd, buf, count, op, stream):
            try:
                if hip.hipStreamSynchronize(stream) != 0:
                    return 1
                raw = torch.empty(count, dtype=torch.int32)
                if hip.hipMemcpy(raw.data_ptr(), buf, count * 4, D2H) != 0:
                    return 1
                if op == 0:                                       # TNSX_SLAB_SUM_U32 (the sums stay far below 2^63)
                    v = (raw.to(torch.int64) & 0xffffffff)
                    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
                    raw = torch.from_numpy((v.numpy() & 0xffffffff).astype(np.uint32).view(np.int32).copy())
                else:
                    v = raw.view(torch.float32).clone()
                    dist.all_reduce(v, op=dist.ReduceOp.MIN if op == 1 else dist.ReduceOp.MAX, group=group)
                    raw = v.view(torch.int32)
                return 0 if hip.hipMemcpy(buf, raw.data_ptr(), count * 4, H2D) == 0 else 1
            except Exception:                                     # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(A.SlabOp), C.c_int, C.c_void_p)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p)
        self._cb = (EX(exchange), AR(allreduce))                  # (kept alive with the object)
        self.t.user = None
        self.t.exchange = C.cast(self._cb[0], C.c_void_p)
        self.t.allreduce = C.cast(self._cb[1], C.c_void_p)
        self.t.release = None
        self.t.abort = None
        self.kind = "host"
        return self

    @staticmethod
    def local_group(world: int):
        import ctypes as C
        from . import api as A
        h = C.c_void_p()
        if A.load_library().tnsx_slab_local_group_create(int(world), C.byref(h)) != 0:
            raise RuntimeError("tnsx_slab_local_group_create failed")
        return h

    @staticmethod
    def local_group_release(h) -> None:
        from . import api as A
        A.load_library().tnsx_slab_local_group_release(h)

    def release(self) -> None:
        import ctypes as C
        if self.kind is not None and self.kind != "host":
            self._L.tnsx_slab_transport_release(C.byref(self.t))
        self.kind = None

    def __del__(self):
        # (the RCCL communicator / the local group's reference must not leak when release() is never called by hand)
        try:
            self.release()
        except Exception:
            pass


def balanced_cuts_c(engine, transport: Optional[SlabTransportC], rank: int, world: int, point_sets: Sequence[torch.Tensor], plane_width: float,
                    n_slabs: Optional[int] = None) -> np.ndarray:
    """tnsx_slab_balanced_cuts: collective over the transport; -> float32[n_slabs + 1] with -inf / +inf at the ends."""
    import ctypes as C
    L = engine._L
    n = len(point_sets)
    ns = int(n_slabs) if n_slabs else int(world)
    ptrs = (C.c_void_p * n)(*[p.data_ptr() if p.shape[0] else None for p in point_sets])
    cnts = (C.c_int * n)(*[int(p.shape[0]) for p in point_sets])
    out = (C.c_float * (ns + 1))()
    engine._wait_for_producers_of(point_sets)
    st = L.tnsx_slab_balanced_cuts(engine._h, C.byref(transport.t) if transport is not None else None, int(rank), int(world), n, ptrs, cnts,
                                   C.c_float(float(plane_width)), ns, out)
    if st != 0:
        raise ValueError(f"tnsx_slab_balanced_cuts failed with status {st}: " + (L.tnsx_slab_last_error(None) or b"").decode())
    return np.array(list(out), np.float32)


def redistribute_c(engine, transport: Optional["SlabTransportC"], rank: int, world: int, cuts, pts: torch.Tensor, gids: torch.Tensor,
                   radii: Optional[torch.Tensor] = None):
    """tnsx_slab_redistribute_begin / _finish: the all-to-all that moves every point to the slab owning its x (collective over the
    transport).  -> (points, ids[, radii]) of the points this rank owns, as new tensors on the engine's device."""
    import ctypes as C
    L = engine._L
    assert pts.is_cuda and pts.dtype == torch.float32 and pts.is_contiguous() and gids.dtype == torch.int64 and gids.is_contiguous()
    assert radii is None or (radii.is_cuda and radii.dtype == torch.float32 and radii.is_contiguous())
    cc = (C.c_float * (world + 1))(*[float(x) for x in cuts])
    h, n_owned = C.c_void_p(), C.c_int(0)
    engine._wait_for_producers_of([pts, gids] + ([radii] if radii is not None else []))
    n = int(pts.shape[0])
    st = L.tnsx_slab_redistribute_begin(engine._h, C.byref(transport.t) if transport is not None else None, int(rank), int(world), cc,
                                        pts.data_ptr() if n else None, gids.data_ptr() if n else None, radii.data_ptr() if (radii is not None and n) else None, n,
                                        C.byref(h), C.byref(n_owned))
    if st != 0:
        raise RuntimeError(f"tnsx_slab_redistribute_begin failed with status {st}: " + (L.tnsx_slab_last_error(None) or b"").decode())
    m = int(n_owned.value)
    o_pts = torch.empty((max(m, 1), 3), dtype=torch.float32, device=pts.device)[:m]
    o_gid = torch.empty(max(m, 1), dtype=torch.int64, device=pts.device)[:m]
    o_rad = torch.empty(max(m, 1), dtype=torch.float32, device=pts.device)[:m] if radii is not None else None
    torch.cuda.current_stream(pts.device).synchronize()   # (the outputs were allocated on torch's stream; the library writes them on the engine's)
    st = L.tnsx_slab_redistribute_finish(h, o_pts.data_ptr(), o_gid.data_ptr(), o_rad.data_ptr() if o_rad is not None else None)
    if st != 0:
        raise RuntimeError(f"tnsx_slab_redistribute_finish failed with status {st}: " + (L.tnsx_slab_last_error(None) or b"").decode())
    return (o_pts, o_gid) if radii is None else (o_pts, o_gid, o_rad)


def transport_check_c(engine, transport: Optional["SlabTransportC"], rank: int, world: int) -> int:
    """tnsx_slab_transport_check: all-reduces the number 1 over the transport's ranks (collective) -> how many ranks are on the other end."""
    import ctypes as C
    seen = C.c_int(0)
    st = engine._L.tnsx_slab_transport_check(engine._h, C.byref(transport.t) if transport is not None else None, int(rank), int(world), C.byref(seen))
    if st != 0:
        raise RuntimeError(f"tnsx_slab_transport_check failed with status {st}")
    return int(seen.value)


class SlabSearchC:
    """SlabSearch on the C entry points (tnsx_slab_create / tnsx_slab_step).  Same calling convention as SlabSearch:
    step(pts, gids[, radii]) or step((pts, gids[, radii]), ...); the lists are read from `engine` with the set ids of `set_id(k)`."""

    def __init__(self, slab_lo: float, slab_hi: float, radius: Optional[float], engine, transport: Optional[SlabTransportC], rank: int, world: int,
                 halo_margin: float = 1.0e-3, max_radius: Optional[float] = None, speculative: bool = True):
        import ctypes as C
        self.engine = engine
        self._L = engine._L
        self.transport = transport
        self.rank, self.world = int(rank), int(world)
        self.variable = radius is None
        if self.variable and max_radius is None:
            raise ValueError("per-point radii: max_radius (an upper bound of every search radius) is needed to size the halo")
        h = C.c_void_p()
        st = self._L.tnsx_slab_create(engine._h, C.byref(transport.t) if transport is not None else None, self.rank, self.world, C.c_float(float(slab_lo)),
                                      C.c_float(float(slab_hi)), C.c_float(-1.0 if self.variable else float(radius)),
                                      C.c_float(float(max_radius) if self.variable else float(radius)), C.c_float(float(halo_margin)), int(bool(speculative)), C.byref(h))
        if st != 0:
            raise RuntimeError(f"tnsx_slab_create failed with status {st}: " + (self._L.tnsx_slab_last_error(None) or b"").decode())
        self._h = h
        self._keep = None
        self.n_sets = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.tnsx_slab_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_symmetric_search(self, active: bool) -> None:
        self.engine.set_symmetric_search(active)

    def set_watchdog(self, seconds: float) -> None:
        """every wait of step() on the stream is bounded by `seconds` (default 120; <= 0: for ever): TNSX_ERR_TIMEOUT names the link"""
        import ctypes as C
        self._check(self._L.tnsx_slab_set_watchdog(self._h, C.c_double(float(seconds))))

    def set_active_search(self, i: int, j: int, active: bool = True) -> None:
        self._check(self._L.tnsx_slab_set_active_search(self._h, int(i), int(j), int(bool(active))))

    def set_collect_times(self, on: bool) -> None:
        """hipEvent pairs around the exchange rounds of every step from now on (info().exchange_ms_last); off in timed loops"""
        self._check(self._L.tnsx_slab_set_collect_times(self._h, int(bool(on))))

    def _check(self, st):
        if st != 0:
            raise RuntimeError(f"slab layer error {st}: " + (self._L.tnsx_slab_last_error(self._h) or b"").decode())

    def step(self, *sets):
        import ctypes as C
        if sets and torch.is_tensor(sets[0]):
            sets = (tuple(sets),)
        sets = [tuple(s) + (None,) * (3 - len(s)) for s in sets]
        n = len(sets)
        for (p, g, r) in sets:
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.dtype == torch.int64 and g.is_contiguous()
            assert (r is not None) == self.variable, "per-point radii must be given for every set, or for none (fixed radius)"
        xyz = (C.c_void_p * n)(*[p.data_ptr() if p.shape[0] else None for (p, g, r) in sets])
        gid = (C.c_void_p * n)(*[g.data_ptr() if g.shape[0] else None for (p, g, r) in sets])
        rad = (C.c_void_p * n)(*[(r.data_ptr() if (r is not None and r.shape[0]) else None) for (p, g, r) in sets])
        cnt = (C.c_int * n)(*[int(p.shape[0]) for (p, g, r) in sets])
        self._keep = sets
        self.n_sets = n
        self.engine._views = {}
        self.engine._wait_for_producers_of([t for s in sets for t in s if t is not None])
        self._check(self._L.tnsx_slab_step(self._h, n, xyz, gid, rad if self.variable else None, cnt))

    def set_id(self, k: int = 0) -> int:
        return int(self._L.tnsx_slab_engine_set(self._h, int(k)))

    def info(self):
        import ctypes as C
        from . import api as A
        inf = A.SlabInfo()
        self._check(self._L.tnsx_slab_get_info(self._h, C.byref(inf)))
        return inf

    @property
    def n_owned(self) -> int:
        return int(self.info().n_owned)

    def neighbors_device(self, i: int = 0, j: int = 0):
        return self.engine.pair_view(self.set_id(i), self.set_id(j))

    def debug_set_capacity(self, side: int, rows: int) -> None:
        self._check(self._L.tnsx_slab_debug_set_capacity(self._h, int(side), int(rows)))
