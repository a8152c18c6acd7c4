"""Spatial-slab sharding of one neighbour search across the GPUs of a node.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI; "gloo" on CPU for the tests).  Rank k owns
the points of slab k along x.  Per step there is exactly ONE exchange: every rank sends the points lying within one
halo width (>= the search radius) of its left / right slab face to that neighbour (grouped isend/irecv of a count,
then xyz(+r) payload and the global ids).  The received ghosts become a second point set, and the engine runs the
pairs (owned -> owned) and (owned -> ghost) -- the multi-set machinery of the reference API is exactly what a halo
needs, so the single-GPU engine is used unchanged and no collective touches the data path.

Results: for every owned point two lists, indices local to `owned` resp. to `ghost`; `global_neighbors()` translates
both to global ids, which makes the union identical to the single-device result on the union of all slabs.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist


def slab_halo_masks(x: torch.Tensor, lo: float, hi: float, halo: float, has_left: bool, has_right: bool):
    """Boolean masks of the owned points that the left / right neighbour needs as ghosts."""
    left = (x < (lo + halo)) if has_left else torch.zeros_like(x, dtype=torch.bool)
    right = (x >= (hi - halo)) if has_right else torch.zeros_like(x, dtype=torch.bool)
    return left, right


class SlabExchange:
    """Ghost-halo exchange between neighbouring slabs.  Works on CPU tensors with gloo and CUDA tensors with RCCL."""

    def __init__(self, slab_lo: float, slab_hi: float, halo: float, group=None):
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.lo, self.hi, self.halo = float(slab_lo), float(slab_hi), float(halo)
        self.bytes_sent = 0

    def exchange(self, pts: torch.Tensor, gids: torch.Tensor, radii: Optional[torch.Tensor] = None):
        """pts (n,3) float32, gids (n,) int64 global ids, radii (n,) float32 or None.
        -> (ghost_pts (m,3), ghost_gids (m,), ghost_radii (m,) or None)"""
        dev = pts.device
        has_l, has_r = self.rank > 0, self.rank < self.world - 1
        ml, mr = slab_halo_masks(pts[:, 0], self.lo, self.hi, self.halo, has_l, has_r)
        cols = 4 if radii is None else 5
        # payload rows: x, y, z, [r], and the 64-bit global id bit-cast into two float32 columns
        def pack(mask):
            sel = torch.nonzero(mask, as_tuple=False).squeeze(1)
            out = torch.empty((sel.numel(), cols + 1), dtype=torch.float32, device=dev)
            out[:, 0:3] = pts.index_select(0, sel)
            if radii is not None:
                out[:, 3] = radii.index_select(0, sel)
            out[:, cols - 1:cols + 1] = gids.index_select(0, sel).view(-1, 1).view(torch.float32).view(-1, 2)
            return out

        send = {}
        if has_l:
            send[self.rank - 1] = pack(ml)
        if has_r:
            send[self.rank + 1] = pack(mr)
        peers = sorted(send.keys())
        # 1) counts
        cnt_out = {p: torch.tensor([send[p].shape[0]], dtype=torch.int64, device=dev) for p in peers}
        cnt_in = {p: torch.zeros(1, dtype=torch.int64, device=dev) for p in peers}
        ops = []
        for p in peers:
            ops.append(dist.P2POp(dist.isend, cnt_out[p], p, self.group))
            ops.append(dist.P2POp(dist.irecv, cnt_in[p], p, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        # 2) payload
        recv = {p: torch.empty((int(cnt_in[p].item()), cols + 1), dtype=torch.float32, device=dev) for p in peers}
        ops = []
        for p in peers:
            if send[p].numel():
                ops.append(dist.P2POp(dist.isend, send[p], p, self.group))
                self.bytes_sent += send[p].numel() * 4
            if recv[p].numel():
                ops.append(dist.P2POp(dist.irecv, recv[p], p, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if peers:
            g = torch.cat([recv[p] for p in peers], dim=0)
        else:
            g = torch.empty((0, cols + 1), dtype=torch.float32, device=dev)
        ghost_pts = g[:, 0:3].contiguous()
        ghost_r = g[:, 3].contiguous() if radii is not None else None
        ghost_gid = g[:, cols - 1:cols + 1].contiguous().view(torch.int64).view(-1)
        return ghost_pts, ghost_gid, ghost_r


class SlabSearch:
    """Owned + ghost search of one slab.  `engine_factory()` must return an object with the TreeNSearch API
    (treensearch_amd.TreeNSearch on a GPU; the CPU tests inject an oracle-backed stand-in)."""

    def __init__(self, slab_lo: float, slab_hi: float, radius: float, engine_factory: Callable[[], object],
                 halo_margin: float = 1.0e-3, group=None):
        self.radius = float(radius)
        self.ex = SlabExchange(slab_lo, slab_hi, self.radius * (1.0 + halo_margin), group)
        self.engine = engine_factory()
        self.engine.set_search_radius(radius)
        self._owned_set = None
        self._ghost_set = None
        self.ghost_gids = None
        self.owned_gids = None
        self._ghost_buf = None

    def step(self, pts: torch.Tensor, gids: torch.Tensor):
        """One exchange + one run.  pts must stay alive (the engine keeps the pointer like the reference does)."""
        ghost_pts, ghost_gid, _ = self.ex.exchange(pts, gids)
        self.owned_gids, self.ghost_gids = gids, ghost_gid
        self._ghost_buf = ghost_pts
        e = self.engine
        if self._owned_set is None:
            self._owned_set = e.add_point_set(pts)
            self._ghost_set = e.add_point_set(ghost_pts)
            e.set_active_search(self._owned_set, self._owned_set, True)
            e.set_active_search(self._owned_set, self._ghost_set, True)
        else:
            e.resize_point_set(self._owned_set, pts)
            e.resize_point_set(self._ghost_set, ghost_pts)
        e.run()

    def global_neighbors(self):
        """(offsets int64[n+1], global ids int64[E]) of the owned points, every list ascending."""
        e = self.engine
        o0, i0 = e.neighbor_csr(self._owned_set, self._owned_set)
        o1, i1 = e.neighbor_csr(self._owned_set, self._ghost_set)
        og = self.owned_gids.cpu().numpy()
        gg = self.ghost_gids.cpu().numpy()
        n = len(o0) - 1
        c0, c1 = np.diff(o0), np.diff(o1)
        offs = np.zeros(n + 1, np.int64)
        np.cumsum(c0 + c1, out=offs[1:])
        out = np.empty(int(offs[-1]), np.int64)
        # interleave the two lists per point
        pos0 = np.repeat(offs[:-1] - o0[:-1], c0) + np.arange(len(i0))
        pos1 = np.repeat(offs[:-1] + c0 - o1[:-1], c1) + np.arange(len(i1))
        out[pos0] = og[i0]
        out[pos1] = gg[i1] if len(gg) else np.zeros(0, np.int64)
        lid = np.repeat(np.arange(n), c0 + c1)
        order = np.lexsort((out, lid))
        return offs, out[order]
