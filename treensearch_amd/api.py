"""ctypes mirror of the reference's `tns::TreeNSearch` class on top of the C ABI (include/tnsx.h).

Same method names, argument meaning and defaults as /root/reference/TreeNSearch/source/TreeNSearch.h:28-427
(all searches inactive by default, symmetric search on, fixed radius XOR per-point radii, set-local indices,
cell size write-once).  Where the reference prints a message and calls exit(-1) this mirror raises TnsxError
with the same message.  There is NO CPU fallback: the native library must load and a gfx950 device must exist.

Point data may be numpy arrays (host memory, re-read at every run() exactly like the reference re-reads the
user's raw pointers, TreeNSearch.h:375-378) or torch CUDA tensors (HBM, read in place).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TNSX_LIB") or os.path.join(_PKG, "lib", "libtnsx.so")   # TNSX_LIB: A/B builds of the same ABI

ARITH_STRICT = 0
ARITH_CONTRACTED = 1

TNSX_F32, TNSX_F64, TNSX_HOST, TNSX_DEVICE, TNSX_VARIABLE = 0, 1, 0, 2, 4


class TnsxError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[tnsx status {status}] {message}")
        self.status = status
        self.message = message


class _Options(C.Structure):
    _fields_ = [("device_id", C.c_int), ("stream", C.c_void_p), ("arith", C.c_int), ("mirror_to_host", C.c_int),
                ("collect_stage_times", C.c_int), ("exact_layout", C.c_int), ("max_dense_cells", C.c_uint64), ("temporal_reuse", C.c_int),
                ("sorted_lists", C.c_int), ("n_devices", C.c_int), ("device_ids", C.c_int * 8), ("query_blocks_per_cu", C.c_int),
                ("fast_blocks_per_cu", C.c_int), ("bucket_build_min_points", C.c_int), ("sparse_grid", C.c_int), ("query_formulation", C.c_int)]


class _CsrView(C.Structure):
    _fields_ = [("n_points", C.c_int), ("n_records", C.c_uint64), ("n_neighbors", C.c_uint64),
                ("offsets_device", C.c_void_p), ("records_device", C.c_void_p),
                ("offsets_host", C.c_void_p), ("records_host", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("n_sets", C.c_int), ("n_points", C.c_uint64), ("n_queries", C.c_uint64), ("n_neighbors", C.c_uint64),
                ("n_occupied_cells", C.c_uint64), ("n_grid_cells", C.c_uint64), ("grid_dims", C.c_int * 3),
                ("grid_cell_size", C.c_float), ("grid_origin", C.c_float * 3), ("key_bits", C.c_int), ("radix_passes", C.c_int),
                ("bytes_build", C.c_uint64), ("bytes_query", C.c_uint64),
                ("ms_total", C.c_float), ("ms_upload", C.c_float), ("ms_bounds", C.c_float), ("ms_table_clear", C.c_float),
                ("ms_sort", C.c_float), ("ms_cells", C.c_float), ("ms_count", C.c_float),
                ("ms_scan", C.c_float), ("ms_fill", C.c_float), ("ms_mirror", C.c_float), ("ms_sort_lists", C.c_float),
                ("n_pool_pairs", C.c_int), ("pool_retries", C.c_int), ("cold_passes", C.c_int), ("speculated", C.c_int),
                ("speculation_redos", C.c_int), ("n_cached_sets", C.c_int), ("n_filtered_cells", C.c_uint32), ("n_devices_used", C.c_int),
                ("world_bottom", C.c_float * 3), ("world_top", C.c_float * 3), ("world_cells_pow2", C.c_int), ("zsort_cell_size_inv", C.c_float),
                ("grid_trimmed", C.c_int), ("n_group_pairs", C.c_uint32), ("n_group_passed_cells", C.c_uint32), ("grid_sparse", C.c_int), ("one_read_builds", C.c_int), ("heavy_catchups", C.c_int), ("sampled_passes", C.c_int), ("nan_fixups", C.c_int)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


# every symbol include/tnsx.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = [
    "tnsx_default_options", "tnsx_create", "tnsx_destroy", "tnsx_last_error", "tnsx_version", "tnsx_get_device", "tnsx_query_formulation_available",
    "tnsx_add_point_set", "tnsx_resize_point_set",
    "tnsx_set_search_radius", "tnsx_set_cell_size", "tnsx_set_symmetric_search", "tnsx_set_active_search",
    "tnsx_set_active_search_all", "tnsx_set_all_searches", "tnsx_set_arithmetic", "tnsx_set_collect_stage_times",
    "tnsx_get_n_sets", "tnsx_get_n_points_in_set", "tnsx_get_total_n_points", "tnsx_is_search_active",
    "tnsx_does_set_exist", "tnsx_get_neighborlist_n_bytes",
    "tnsx_run", "tnsx_run_scalar", "tnsx_get_pair_view", "tnsx_mirror_pair_to_host", "tnsx_copy_pair", "tnsx_pair_csr_device",
    "tnsx_prepare_zsort", "tnsx_get_zsort_order", "tnsx_apply_zsort", "tnsx_get_stats",
    "tnsx_halo_pack", "tnsx_x_histogram", "tnsx_set_query_count", "tnsx_translate_neighbors", "tnsx_set_point_ids", "tnsx_synchronize",
    # slab layer (tnsx_slab.cpp)
    "tnsx_slab_rccl_unique_id", "tnsx_slab_transport_rccl", "tnsx_slab_rccl_error", "tnsx_slab_local_group_create", "tnsx_slab_local_group_release",
    "tnsx_slab_transport_local", "tnsx_slab_transport_release", "tnsx_slab_balanced_cuts", "tnsx_slab_create", "tnsx_slab_destroy",
    "tnsx_slab_last_error", "tnsx_slab_set_active_search", "tnsx_slab_step", "tnsx_slab_engine_set", "tnsx_slab_get_info",
    "tnsx_slab_debug_set_capacity", "tnsx_slab_set_watchdog", "tnsx_slab_redistribute_begin", "tnsx_slab_redistribute_finish",
    "tnsx_slab_set_collect_times", "tnsx_slab_transport_check", "tnsx_slab_set_redistribute_watchdog",
]


class SlabTransport(C.Structure):
    """tnsx_slab_transport (include/tnsx.h): filled by tnsx_slab_transport_rccl / tnsx_slab_transport_local"""
    _fields_ = [("user", C.c_void_p), ("exchange", C.c_void_p), ("allreduce", C.c_void_p), ("release", C.c_void_p), ("abort", C.c_void_p)]


class SlabOp(C.Structure):
    """tnsx_slab_op: one message pair with one neighbour (a transport's exchange gets an array of them)"""
    _fields_ = [("peer", C.c_int), ("send", C.c_void_p), ("send_bytes", C.c_size_t), ("recv", C.c_void_p), ("recv_bytes", C.c_size_t)]


class SlabInfo(C.Structure):
    _fields_ = [("n_owned", C.c_int), ("n_ghost", C.c_int), ("speculative_last", C.c_int), ("redone_last", C.c_int), ("rounds_last", C.c_int),
                ("bytes_sent", C.c_ulonglong), ("transport_kind", C.c_int), ("transport_ranks", C.c_int), ("exchange_ms_last", C.c_float)]

_lib = None


def load_library():
    """Loads libtnsx.so (built by treensearch_amd.build).  Raises if it is missing -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found: build it with `python -m treensearch_amd.build` "
                      "(the engine has no CPU fallback)")
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in one process do not
    # share devices or pointers, so torch's copy must be mapped BEFORE libtnsx.so: the loader then resolves our
    # DT_NEEDED libamdhip64.so.7 to the already-loaded one and torch tensors / streams are valid in the engine.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    L.tnsx_default_options.argtypes = [C.POINTER(_Options)]
    L.tnsx_create.argtypes = [C.POINTER(_Options), C.POINTER(vp)]
    L.tnsx_destroy.argtypes = [vp]
    L.tnsx_destroy.restype = None
    L.tnsx_last_error.argtypes = [vp]
    L.tnsx_last_error.restype = C.c_char_p
    L.tnsx_query_formulation_available.argtypes = [ci]
    L.tnsx_halo_pack.argtypes = [vp, vp, vp, vp, ci, C.c_float, C.c_float, vp, vp, C.c_ulonglong, C.c_ulonglong, vp, C.POINTER(C.c_uint)]
    L.tnsx_x_histogram.argtypes = [vp, vp, ci, C.c_float, C.c_float, ci, vp]
    L.tnsx_set_query_count.argtypes = [vp, ci, ci]
    L.tnsx_translate_neighbors.argtypes = [vp, ci, ci, vp]
    L.tnsx_set_point_ids.argtypes = [vp, ci, vp]
    L.tnsx_synchronize.argtypes = [vp]
    L.tnsx_add_point_set.argtypes = [vp, vp, vp, ci, C.c_uint]
    L.tnsx_resize_point_set.argtypes = [vp, ci, vp, vp, ci, C.c_uint]
    L.tnsx_set_search_radius.argtypes = [vp, C.c_float]
    L.tnsx_set_cell_size.argtypes = [vp, C.c_float]
    L.tnsx_set_symmetric_search.argtypes = [vp, ci]
    L.tnsx_set_active_search.argtypes = [vp, ci, ci, ci]
    L.tnsx_set_active_search_all.argtypes = [vp, ci, ci, ci]
    L.tnsx_set_all_searches.argtypes = [vp, ci]
    L.tnsx_set_arithmetic.argtypes = [vp, ci]
    L.tnsx_set_collect_stage_times.argtypes = [vp, ci]
    L.tnsx_get_n_sets.argtypes = [vp]
    L.tnsx_get_n_points_in_set.argtypes = [vp, ci]
    L.tnsx_get_total_n_points.argtypes = [vp]
    L.tnsx_get_total_n_points.restype = C.c_int64
    L.tnsx_is_search_active.argtypes = [vp, ci, ci]
    L.tnsx_does_set_exist.argtypes = [vp, ci]
    L.tnsx_get_neighborlist_n_bytes.argtypes = [vp]
    L.tnsx_get_neighborlist_n_bytes.restype = C.c_uint64
    L.tnsx_run.argtypes = [vp]
    L.tnsx_run_scalar.argtypes = [vp]
    L.tnsx_get_pair_view.argtypes = [vp, ci, ci, C.POINTER(_CsrView)]
    L.tnsx_mirror_pair_to_host.argtypes = [vp, ci, ci]
    L.tnsx_copy_pair.argtypes = [vp, ci, ci, vp, vp, ci]
    L.tnsx_pair_csr_device.argtypes = [vp, ci, ci, vp, vp]
    L.tnsx_get_device.argtypes = [vp]
    L.tnsx_prepare_zsort.argtypes = [vp]
    L.tnsx_get_zsort_order.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(ci)]
    L.tnsx_apply_zsort.argtypes = [vp, ci, vp, C.c_size_t, ci, ci]
    L.tnsx_get_stats.argtypes = [vp, C.POINTER(Stats)]
    # slab layer
    tp = C.POINTER(SlabTransport)
    L.tnsx_slab_rccl_unique_id.argtypes = [vp]
    L.tnsx_slab_transport_rccl.argtypes = [vp, ci, ci, ci, tp]
    L.tnsx_slab_rccl_error.restype = C.c_char_p
    L.tnsx_slab_local_group_create.argtypes = [ci, C.POINTER(vp)]
    L.tnsx_slab_local_group_release.argtypes = [vp]
    L.tnsx_slab_local_group_release.restype = None
    L.tnsx_slab_transport_local.argtypes = [vp, ci, tp]
    L.tnsx_slab_transport_release.argtypes = [tp]
    L.tnsx_slab_transport_release.restype = None
    L.tnsx_slab_balanced_cuts.argtypes = [vp, tp, ci, ci, ci, C.POINTER(vp), C.POINTER(ci), C.c_float, ci, C.POINTER(C.c_float)]
    L.tnsx_slab_create.argtypes = [vp, tp, ci, ci, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, ci, C.POINTER(vp)]
    L.tnsx_slab_destroy.argtypes = [vp]
    L.tnsx_slab_destroy.restype = None
    L.tnsx_slab_last_error.argtypes = [vp]
    L.tnsx_slab_last_error.restype = C.c_char_p
    L.tnsx_slab_set_active_search.argtypes = [vp, ci, ci, ci]
    L.tnsx_slab_step.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(ci)]
    L.tnsx_slab_engine_set.argtypes = [vp, ci]
    L.tnsx_slab_get_info.argtypes = [vp, C.POINTER(SlabInfo)]
    L.tnsx_slab_debug_set_capacity.argtypes = [vp, ci, C.c_uint]
    L.tnsx_slab_set_watchdog.argtypes = [vp, C.c_double]
    L.tnsx_slab_set_collect_times.argtypes = [vp, ci]
    L.tnsx_slab_set_redistribute_watchdog.argtypes = [C.c_double]
    L.tnsx_slab_transport_check.argtypes = [vp, tp, ci, ci, C.POINTER(ci)]
    L.tnsx_slab_redistribute_begin.argtypes = [vp, tp, ci, ci, C.POINTER(C.c_float), vp, vp, vp, ci, C.POINTER(vp), C.POINTER(ci)]
    L.tnsx_slab_redistribute_finish.argtypes = [vp, vp, vp, vp]
    _lib = L
    return L


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _describe(arr, want_cols: Optional[int]):
    """-> (pointer, n_elements, flags, keepalive)"""
    if _is_torch(arr):
        import torch
        if arr.dtype not in (torch.float32, torch.float64):
            raise TypeError("points/radii must be float32 or float64")
        if not arr.is_contiguous():
            raise ValueError("tensor must be contiguous (xyzxyz... layout)")
        flags = (TNSX_F64 if arr.dtype == torch.float64 else TNSX_F32) | (TNSX_DEVICE if arr.is_cuda else TNSX_HOST)
        return arr.data_ptr() if arr.numel() else None, arr.numel(), flags, arr
    a = np.asarray(arr)
    if a.dtype not in (np.float32, np.float64):
        raise TypeError("points/radii must be float32 or float64")
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous (xyzxyz... layout)")
    flags = (TNSX_F64 if a.dtype == np.float64 else TNSX_F32) | TNSX_HOST
    return a.ctypes.data if a.size else None, a.size, flags, a


class _DeviceArray:
    """A raw device pointer dressed as a CUDA array (the __cuda_array_interface__ protocol, version 2) so that torch.as_tensor wraps it without a
    copy; keeps the owner of the memory alive."""

    def __init__(self, ptr: int, n: int, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}
        self._owner = owner


def _device_view(ptr, n: int, typestr: str, itemsize: int, device, owner):
    import torch
    dtype = {"<i8": torch.int64, "<i4": torch.int32}[typestr]
    if n == 0 or not ptr:
        return torch.empty(0, dtype=dtype, device=device)
    p = ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value
    t = torch.as_tensor(_DeviceArray(p, n, typestr, owner), device=device)
    assert t.data_ptr() == p and t.dtype == dtype, "torch copied the array instead of wrapping it"
    return t


class NeighborList:
    """tns::NeighborList (NeighborList.h:8-39): a view of one `[count, j0, j1, ...]` record."""

    __slots__ = ("_rec",)

    def __init__(self, rec: np.ndarray):
        self._rec = rec

    def size(self) -> int:
        return int(self._rec[0])

    def __len__(self):
        return int(self._rec[0])

    def __getitem__(self, i):
        return int(self._rec[1 + i])

    def get_ptr(self) -> np.ndarray:
        return self._rec[1:1 + int(self._rec[0])]


class TreeNSearch:
    def __init__(self, *, arith: int = ARITH_STRICT, mirror_to_host: bool = False, device_id: int = -1,
                 stream: Optional[int] = None, collect_stage_times: bool = False, max_dense_cells: int = 0,
                 exact_layout: bool = False, temporal_reuse: bool = True, sorted_lists: bool = False, devices=None,
                 query_blocks_per_cu: int = 0, fast_blocks_per_cu: int = 0, bucket_build_min_points: int = 0,
                 query_formulation: int = 0, sparse_grid: int = 0):
        """devices: a list of HIP device ordinals -> multi-device mode (host-resident inputs only, see include/tnsx.h)"""
        self._L = load_library()
        opt = _Options()
        self._L.tnsx_default_options(C.byref(opt))
        opt.device_id = device_id
        opt.stream = stream
        opt.arith = arith
        opt.mirror_to_host = int(mirror_to_host)
        opt.collect_stage_times = int(collect_stage_times)
        opt.max_dense_cells = max_dense_cells
        opt.exact_layout = int(exact_layout)
        opt.temporal_reuse = int(temporal_reuse)
        opt.sorted_lists = int(sorted_lists)
        opt.query_blocks_per_cu = int(query_blocks_per_cu)
        opt.fast_blocks_per_cu = int(fast_blocks_per_cu)
        opt.bucket_build_min_points = int(bucket_build_min_points)
        opt.sparse_grid = int(sparse_grid)
        opt.query_formulation = int(query_formulation)
        if devices is not None and len(devices) > 1:
            opt.n_devices = len(devices)
            for k, d in enumerate(devices):
                opt.device_ids[k] = int(d)
        h = C.c_void_p()
        st = self._L.tnsx_create(C.byref(opt), C.byref(h))
        if st != 0:
            raise TnsxError(st, self._L.tnsx_last_error(None).decode())
        self._h = h
        self._keep = {}
        self._views = {}
        self._n_threads = -1
        self._own_stream = not stream          # the engine then runs on a stream of its own (non-blocking)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.tnsx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _check(self, st: int):
        if st != 0:
            raise TnsxError(st, self._L.tnsx_last_error(self._h).decode())

    def _set_args(self, points, radii, n_points):
        pp, pn, pf, pk = _describe(points, 3)
        if radii is not None:
            rp, rn, rf, rk = _describe(radii, 1)
            if (rf & TNSX_F64) != (pf & TNSX_F64) or (rf & TNSX_DEVICE) != (pf & TNSX_DEVICE):
                raise TypeError("points and radii must have the same dtype and live in the same memory")
            flags = pf | TNSX_VARIABLE
        else:
            rp, rn, rk = None, 0, None
            flags = pf
        n = pn // 3 if n_points is None else int(n_points)
        if n * 3 > pn or (radii is not None and n > rn):
            raise ValueError("n_points exceeds the array size")
        return pp, rp, n, flags, (pk, rk)

    # ------------------------------------------------------------------ main interface
    def add_point_set(self, points, radii=None, n_points: Optional[int] = None) -> int:
        """TreeNSearch.h:50,63,112,126.  `radii is None` => fixed-radius mode."""
        pp, rp, n, flags, keep = self._set_args(points, radii, n_points)
        s = self._L.tnsx_add_point_set(self._h, pp, rp, n, flags)
        if s < 0:
            self._check(-s)
        self._keep[s] = keep
        return s

    def resize_point_set(self, set_id: int, points, radii=None, n_points: Optional[int] = None) -> None:
        """TreeNSearch.h:72,81,136,146."""
        pp, rp, n, flags, keep = self._set_args(points, radii, n_points)
        self._check(self._L.tnsx_resize_point_set(self._h, int(set_id), pp, rp, n, flags))
        self._keep[set_id] = keep

    def set_search_radius(self, r) -> None:
        self._check(self._L.tnsx_set_search_radius(self._h, C.c_float(float(np.float32(r)))))

    def set_cell_size(self, c) -> None:
        self._check(self._L.tnsx_set_cell_size(self._h, C.c_float(float(np.float32(c)))))

    def set_symmetric_search(self, activate: bool) -> None:
        self._check(self._L.tnsx_set_symmetric_search(self._h, int(bool(activate))))

    def set_arithmetic(self, arith: int) -> None:
        self._check(self._L.tnsx_set_arithmetic(self._h, int(arith)))

    def set_collect_stage_times(self, on: bool) -> None:
        """hipEvents around every stage from the next run on (each one is a bubble between two kernels: off in timed loops)"""
        self._check(self._L.tnsx_set_collect_stage_times(self._h, int(bool(on))))

    def _wait_for_producers(self) -> None:
        """Device inputs are read on the engine's stream.  When that is the engine's own stream, work that torch has queued on
        ITS current stream (position updates, ghost copies, NCCL receives) is not ordered before it -- wait for it.  With a
        caller-supplied stream everything is in stream order and nothing is waited for."""
        if self._own_stream and any(_is_torch(k) and k.is_cuda for pair in self._keep.values() for k in pair if k is not None):
            import torch
            torch.cuda.current_stream().synchronize()

    def _wait_for_producers_of(self, tensors) -> None:
        """the same for tensors the engine does not hold (slab layer: the owned points handed to tnsx_slab_step)"""
        if self._own_stream and any(_is_torch(t) and t.is_cuda for t in tensors):
            import torch
            torch.cuda.current_stream().synchronize()

    def run(self) -> None:
        self._views = {}
        self._wait_for_producers()
        self._check(self._L.tnsx_run(self._h))

    def run_scalar(self) -> None:
        """The reference's scalar twin accumulates in double (TreeNSearch.cpp:2080-2087) and is not a parity
        target; here it is the same GPU path as run(), with the world box of the reference's
        scalar path (the tight box; run() unites it with the origin like the reference's SIMD path does)."""
        self._views = {}
        self._wait_for_producers()
        self._check(self._L.tnsx_run_scalar(self._h))

    # ------------------------------------------------------------------ searches
    def set_all_searches(self, active: bool) -> None:
        self._check(self._L.tnsx_set_all_searches(self._h, int(bool(active))))

    def set_active_search(self, set_i: int, a=True, b=True) -> None:
        """Both reference overloads (TreeNSearch.h:265, :275), resolved like C++ does: an `int` second
        argument selects (set_i, set_j, active=True); a `bool` selects (set_i, search_in_all, be_found_by_all)."""
        if isinstance(a, (bool, np.bool_)):
            self._check(self._L.tnsx_set_active_search_all(self._h, int(set_i), int(a), int(bool(b))))
        else:
            self._check(self._L.tnsx_set_active_search(self._h, int(set_i), int(a), int(bool(b))))

    # ------------------------------------------------------------------ no-op tuning knobs of the CPU design
    def set_n_threads(self, n: int) -> None:
        self._n_threads = int(n)

    def set_recursion_cap(self, cap: int) -> None:
        pass

    def set_n_points_for_parallel_octree(self, n: int = 200000) -> None:
        pass

    # ------------------------------------------------------------------ getters
    def get_n_sets(self) -> int:
        return self._L.tnsx_get_n_sets(self._h)

    def get_n_threads(self) -> int:
        return self._n_threads

    def get_n_points_in_set(self, s: int) -> int:
        return self._L.tnsx_get_n_points_in_set(self._h, int(s))

    def get_total_n_points(self) -> int:
        return int(self._L.tnsx_get_total_n_points(self._h))

    def is_search_active(self, i: int, j: int) -> bool:
        return bool(self._L.tnsx_is_search_active(self._h, int(i), int(j)))

    def does_set_exist(self, s: int) -> bool:
        return bool(self._L.tnsx_does_set_exist(self._h, int(s)))

    def get_neighborlist_n_bytes(self) -> int:
        return int(self._L.tnsx_get_neighborlist_n_bytes(self._h))

    def get_stats(self) -> dict:
        return self.get_stats_raw().as_dict()

    def get_stats_raw(self) -> "Stats":
        """the ctypes mirror of tnsx_stats itself (fields by attribute): for loops that must not spend time building a dict per step"""
        st = Stats()
        self._check(self._L.tnsx_get_stats(self._h, C.byref(st)))
        return st

    # ------------------------------------------------------------------ multi-GPU support
    @staticmethod
    def _dev_ptr(t, dtype, what):
        """data pointer of a contiguous CUDA tensor of the given dtype (None passes through)"""
        if t is None:
            return None
        if not (_is_torch(t) and t.is_cuda and t.is_contiguous() and t.dtype == dtype):
            raise TypeError(f"{what} must be a contiguous CUDA tensor of dtype {dtype}")
        return C.c_void_p(t.data_ptr())

    def _torch_sync_in(self):
        if self._own_stream:
            import torch
            torch.cuda.current_stream().synchronize()

    def halo_pack(self, pts, gids, radii, left_cut, right_cut, out_left, out_right, counts, wait: bool = True):
        """tnsx_halo_pack on device tensors (torch, CUDA): selects the points with x < left_cut / x >= right_cut and appends
        them as rows [x, y, z, (r,) gid_lo, gid_hi] to out_left / out_right (None = side not wanted; each side has its own
        capacity = its number of rows).  counts is a 2-element int32 scratch tensor on the device.  Returns (n_left, n_right), the
        number of selected points per side (may exceed the buffers: then only the first rows were written); with wait=False the
        call only enqueues and returns None (the counts are then in `counts` once the stream gets there)."""
        import torch
        self._torch_sync_in()
        f32 = torch.float32
        host = (C.c_uint * 2)()
        self._check(self._L.tnsx_halo_pack(
            self._h, self._dev_ptr(pts, f32, "pts"), self._dev_ptr(radii, f32, "radii"), self._dev_ptr(gids, torch.int64, "gids"),
            int(pts.shape[0]), float(left_cut), float(right_cut), self._dev_ptr(out_left, f32, "out_left"), self._dev_ptr(out_right, f32, "out_right"),
            0 if out_left is None else int(out_left.shape[0]), 0 if out_right is None else int(out_right.shape[0]),
            self._dev_ptr(counts, torch.int32, "counts"), host if wait else None))
        return (int(host[0]), int(host[1])) if wait else None

    def x_histogram(self, pts, x0: float, inv_dx: float, hist) -> None:
        """tnsx_x_histogram: hist[clamp(trunc((x - x0) * inv_dx), 0, len(hist) - 1)] += 1 for every point of the CUDA tensor pts
        (n,3); hist is a CUDA int32 tensor the caller has zeroed.  Enqueued on the engine's stream."""
        import torch
        self._torch_sync_in()
        self._check(self._L.tnsx_x_histogram(self._h, self._dev_ptr(pts, torch.float32, "pts"), int(pts.shape[0]), float(x0), float(inv_dx),
                                             int(hist.numel()), self._dev_ptr(hist, torch.int32, "hist")))

    def set_query_count(self, set_i: int, n_query: int) -> None:
        """Only the first n_query points of set_i get neighbour lists from the next run() on (-1: all)."""
        self._check(self._L.tnsx_set_query_count(self._h, int(set_i), int(n_query)))

    def set_point_ids(self, set_i: int, ids) -> None:
        """The lists of every pair (* -> set_i) hold ids[j] instead of j from the next run() on (ids: CUDA int32 tensor with one
        entry per point, re-read at every run; None switches back to indices)."""
        import torch
        self._ids_keep = getattr(self, "_ids_keep", {})
        self._ids_keep[int(set_i)] = ids
        self._check(self._L.tnsx_set_point_ids(self._h, int(set_i), self._dev_ptr(ids, torch.int32, "ids")))

    def synchronize(self) -> None:
        """waits for the engine's stream"""
        self._check(self._L.tnsx_synchronize(self._h))

    def order_after_engine(self) -> None:
        """Work queued on torch's current stream from now on runs after what the engine has enqueued so far.  Nothing to do when
        the engine was given that stream; an engine with a stream of its own is waited for."""
        if self._own_stream:
            self.synchronize()

    def translate_neighbors(self, set_i: int, set_j: int, id_map) -> None:
        """Every neighbour index j of pair (set_i -> set_j) becomes id_map[j], in place on the device (id_map: CUDA int32)."""
        import torch
        self._torch_sync_in()
        self._views.pop((set_i, set_j), None)
        self._check(self._L.tnsx_translate_neighbors(self._h, int(set_i), int(set_j), self._dev_ptr(id_map, torch.int32, "id_map")))

    def print_state(self) -> None:
        for k, v in self.get_stats().items():
            print(f"{k}: {v}")

    # ------------------------------------------------------------------ results
    def pair_view(self, i: int, j: int) -> _CsrView:
        v = _CsrView()
        self._check(self._L.tnsx_get_pair_view(self._h, int(i), int(j), C.byref(v)))
        return v

    def neighbor_records(self, i: int, j: int):
        """Host copies of the record storage of pair (i,j): (offsets uint64[n_i] by original point index,
        records int32[n_records]) with records[offsets[p]] = count followed by the neighbour indices."""
        key = (i, j)
        if key not in self._views:
            v = self.pair_view(i, j)
            offs = np.zeros(max(v.n_points, 1), np.uint64)
            recs = np.zeros(max(v.n_records, 1), np.int32)
            self._check(self._L.tnsx_copy_pair(self._h, int(i), int(j), offs.ctypes.data, recs.ctypes.data, 0))
            self._views[key] = (offs[:v.n_points], recs[:v.n_records])
        return self._views[key]

    def neighbor_csr(self, i: int, j: int, sort_each: bool = True):
        """Standard CSR in original point order: (offsets int64[n_i+1], indices int32[E])."""
        offs, recs = self.neighbor_records(i, j)
        n = len(offs)
        if n == 0:
            return np.zeros(1, np.int64), np.zeros(0, np.int32)
        o = offs.astype(np.int64)
        counts = recs[o].astype(np.int64)
        out_offs = np.zeros(n + 1, np.int64)
        np.cumsum(counts, out=out_offs[1:])
        total = int(out_offs[-1])
        src = np.repeat(o + 1 - out_offs[:-1], counts) + np.arange(total, dtype=np.int64)
        idx = recs[src]
        if sort_each and total:
            # sort inside each list: stable sort by (list id, value)
            lid = np.repeat(np.arange(n, dtype=np.int64), counts)
            order = np.lexsort((idx, lid))
            idx = idx[order]
        return out_offs, np.ascontiguousarray(idx, np.int32)

    # ---- device-side consumers ("data_ptr() in, CSR tensors out") -- the lists never leave HBM
    def neighbor_records_torch(self, i: int, j: int):
        """The record storage of pair (i, j) as CUDA tensors WITHOUT a copy: (offsets int64[n_i] by original point index, records int32[n_records]) with
        records[offsets[p]] = count of point p followed by its neighbour indices -- what get_neighborlist (TreeNSearch.cpp:241-249) reads, for kernels on
        the GPU.  The tensors are views of the engine's own memory (tnsx_csr_view.*_device): valid until the next run() of this object, read-only by
        contract; the records keep the pool's layout (holes between the blocks of records: address every list through its offset)."""
        import torch
        v = self.pair_view(i, j)
        if not v.offsets_device or not v.records_device:
            raise TnsxError(4, "neighbor_records_torch: the pair has no device view (multi-device contexts hold host views only)")
        dev = torch.device("cuda", self._device_index())
        offs = _device_view(v.offsets_device, max(int(v.n_points), 0), "<i8", 8, dev, self)
        recs = _device_view(v.records_device, max(int(v.n_records), 0), "<i4", 4, dev, self)
        return offs, recs

    def neighbor_csr_torch(self, i: int, j: int, sort_each: bool = False):
        """Standard gap-free CSR of pair (i, j) in original point order, built on the device (tnsx_pair_csr_device: lengths -> scan -> copy) into
        tensors this call allocates: (offsets int64[n_i + 1], indices int32[E]) on the engine's GPU -- the device-side twin of neighbor_csr().
        sort_each: ascending order inside every list (a device sort of (list, index) keys; off by default: the contract is the set)."""
        import torch
        v = self.pair_view(i, j)
        dev = torch.device("cuda", self._device_index())
        n, e = max(int(v.n_points), 0), int(v.n_neighbors)
        offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
        idx = torch.empty(e, dtype=torch.int32, device=dev)
        torch.cuda.current_stream(dev).synchronize() if self._own_stream else None     # (the allocations' previous users, should the caching allocator recycle them)
        self._check(self._L.tnsx_pair_csr_device(self._h, int(i), int(j), C.c_void_p(offs.data_ptr()), C.c_void_p(idx.data_ptr()) if e else None))
        if sort_each and e:
            lid = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), offs[1:] - offs[:-1])
            idx = idx[torch.argsort(lid * (1 << 31) + idx.to(torch.int64))]
        return offs, idx

    def _device_index(self) -> int:
        if not hasattr(self, "_dev_index"):
            self._dev_index = int(self._L.tnsx_get_device(self._h))
        return self._dev_index

    def get_neighborlist(self, set_i: int, set_j: int, point_i: int) -> NeighborList:
        """TreeNSearch.h:182."""
        offs, recs = self.neighbor_records(set_i, set_j)
        o = int(offs[point_i])
        return NeighborList(recs[o:o + 1 + int(recs[o])])

    def for_each_neighbor(self, set_i: int, set_j: int, i: int, f) -> None:
        """TreeNSearch.h:194-195."""
        nl = self.get_neighborlist(set_i, set_j, i)
        for k in range(nl.size()):
            f(nl[k])

    # ------------------------------------------------------------------ zsort
    def prepare_zsort(self) -> None:
        self._wait_for_producers()
        self._check(self._L.tnsx_prepare_zsort(self._h))

    def get_zsort_order(self, set_i: int) -> np.ndarray:
        hp, dp, n = C.c_void_p(), C.c_void_p(), C.c_int()
        self._check(self._L.tnsx_get_zsort_order(self._h, int(set_i), C.byref(hp), C.byref(dp), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, np.int32)
        return np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_int)), shape=(n.value,)).copy()

    def apply_zsort(self, set_i: int, data, stride: int = 1) -> None:
        """In-place permutation of a user array (TreeNSearch.h:214-215); numpy (host) or torch CUDA tensor."""
        if _is_torch(data):
            if not data.is_contiguous():
                raise ValueError("tensor must be contiguous")
            if data.is_cuda and self._own_stream:
                import torch
                torch.cuda.current_stream().synchronize()
            self._check(self._L.tnsx_apply_zsort(self._h, int(set_i), data.data_ptr(), data.element_size(), int(stride),
                                                 1 if data.is_cuda else 0))
        else:
            if not data.flags["C_CONTIGUOUS"] or not data.flags["WRITEABLE"]:
                raise ValueError("array must be C-contiguous and writeable")
            self._check(self._L.tnsx_apply_zsort(self._h, int(set_i), data.ctypes.data, data.itemsize, int(stride), 0))
