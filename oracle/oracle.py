"""TEST INFRASTRUCTURE ONLY -- ctypes bindings of the CPU oracle and of the real reference.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.

  * `Oracle`            oracle/_build/libtns_oracle.so  (oracle/tns_oracle.c, the CPU restatement)
  * `RefTreeNSearch`    oracle/_ref/libtns_ref[_strict].so (the REAL tns::TreeNSearch, built from
    `RefBruteforce`     /root/reference by oracle/Makefile; absent if never built in this tree)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libtns_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libtns_ref.so")
REF_STRICT_SO = os.path.join(HERE, "_ref", "libtns_ref_strict.so")

STRICT = 0
CONTRACTED = 1

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def build(ref: bool = True) -> None:
    """Compile the oracle (always) and the reference (only when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir(os.environ.get("TNS_REFERENCE_DIR", "/root/reference")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref",
                               "REF=" + os.environ.get("TNS_REFERENCE_DIR", "/root/reference")])


def have_ref() -> bool:
    return os.path.exists(REF_SO) and os.path.exists(REF_STRICT_SO)


class _Result(C.Structure):
    _fields_ = [("n", C.c_int), ("total", C.c_int64), ("offsets", _i64p), ("indices", _i32p)]


class Oracle:
    """The CPU restatement."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        self.lib = L = C.CDLL(ORACLE_SO)
        L.tnso_pair_search.restype = C.POINTER(_Result)
        L.tnso_pair_search.argtypes = [_f32p, _f32p, C.c_int, _f32p, _f32p, C.c_int,
                                       C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.tnso_result_free.argtypes = [C.POINTER(_Result)]
        L.tnso_digest_csr.argtypes = [C.c_int, _i64p, _i32p, C.c_int, _u64p]
        L.tnso_world_box_update.restype = C.c_int
        L.tnso_world_box_update.argtypes = [_f32p, _f32p, C.c_float, _i32p]
        L.tnso_tight_bounds.argtypes = [_f32p, C.c_int, _f32p]
        L.tnso_tight_bounds_simd.argtypes = [_f32p, C.c_int, _f32p]
        L.tnso_zsort_keys.argtypes = [_f32p, C.c_int, _f32p, C.c_float, _u64p]
        L.tnso_zsort_order.argtypes = [_f32p, C.c_int, _f32p, C.c_float, _i32p]
        L.tnso_check_zsort.restype = C.c_int
        L.tnso_check_zsort.argtypes = [_u64p, _i32p, C.c_int]
        L.tnso_morton3.restype = C.c_uint64
        L.tnso_morton3.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.tnso_dist_sq.restype = C.c_float
        L.tnso_dist_sq.argtypes = [_f32p, _f32p, C.c_int]
        L.tnso_num_threads.restype = C.c_int
        L.tnso_remap_csr.argtypes = [C.c_int, _i32p, _i32p, _i64p, _i32p, _i64p, _i32p]

    # -- neighbour search of one (set_i -> set_j) pair -> (offsets int64[n+1], indices int32[total])
    def pair_search(self, xa, xb, *, radius=None, ra=None, rb=None, symmetric=True, same_set=False,
                    mode=STRICT, use_grid=True):
        xa = np.ascontiguousarray(xa, dtype=np.float32).reshape(-1, 3)
        xb = np.ascontiguousarray(xb, dtype=np.float32).reshape(-1, 3)
        if ra is not None:
            ra = np.ascontiguousarray(ra, dtype=np.float32)
            rb = np.ascontiguousarray(rb, dtype=np.float32)
            assert len(ra) == len(xa) and len(rb) == len(xb)
        else:
            assert radius is not None
        res = self.lib.tnso_pair_search(_p(xa, _f32p), _p(ra, _f32p), len(xa), _p(xb, _f32p), _p(rb, _f32p), len(xb),
                                        C.c_float(float(radius) if radius is not None else -1.0),
                                        int(symmetric), int(same_set), int(mode), int(use_grid))
        r = res.contents
        n, total = r.n, r.total
        offsets = np.ctypeslib.as_array(r.offsets, shape=(n + 1,)).copy()
        indices = np.ctypeslib.as_array(r.indices, shape=(max(total, 1),))[:total].copy()
        self.lib.tnso_result_free(res)
        return offsets, indices

    def digest(self, offsets, indices, already_sorted=False):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        if indices.size == 0:
            indices = np.zeros(1, np.int32)
        out = np.zeros(2, np.uint64)
        self.lib.tnso_digest_csr(len(offsets) - 1, _p(offsets, _i64p), _p(indices, _i32p), int(already_sorted),
                                 _p(out, _u64p))
        return int(out[0]), int(out[1])

    def world_box_update(self, box, tight, cell_size):
        """box: float32[6] in/out (bottom, top).  returns (rc, n_cells_pow2)."""
        n = C.c_int(0)
        rc = self.lib.tnso_world_box_update(_p(box, _f32p), _p(np.ascontiguousarray(tight, np.float32), _f32p),
                                            C.c_float(cell_size), C.byref(n))
        return rc, n.value

    def tight_bounds(self, x, tight=None, simd=False):
        """simd=False: run_scalar()'s tight box; simd=True: what run() / prepare_zsort() compute (tight box united with the origin)."""
        if tight is None:
            fm = np.finfo(np.float32).max
            tight = np.array([fm, fm, fm, -fm, -fm, -fm], np.float32)
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 3)
        (self.lib.tnso_tight_bounds_simd if simd else self.lib.tnso_tight_bounds)(_p(x, _f32p), len(x), _p(tight, _f32p))
        return tight

    def zsort_keys(self, x, bottom, cell_size_inv):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 3)
        keys = np.zeros(max(len(x), 1), np.uint64)
        b = np.ascontiguousarray(bottom, np.float32)
        self.lib.tnso_zsort_keys(_p(x, _f32p), len(x), _p(b, _f32p), C.c_float(cell_size_inv), _p(keys, _u64p))
        return keys[:len(x)]

    def zsort_order(self, x, bottom, cell_size_inv):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 3)
        out = np.zeros(max(len(x), 1), np.int32)
        b = np.ascontiguousarray(bottom, np.float32)
        self.lib.tnso_zsort_order(_p(x, _f32p), len(x), _p(b, _f32p), C.c_float(cell_size_inv), _p(out, _i32p))
        return out[:len(x)]

    def check_zsort(self, keys, new_to_old):
        keys = np.ascontiguousarray(keys, np.uint64)
        new_to_old = np.ascontiguousarray(new_to_old, np.int32)
        if len(keys) == 0:
            return 0
        return self.lib.tnso_check_zsort(_p(keys, _u64p), _p(new_to_old, _i32p), len(new_to_old))

    def morton3(self, x, y, z):
        return int(self.lib.tnso_morton3(x, y, z))

    def dist_sq(self, p, q, mode):
        p = np.ascontiguousarray(p, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        return np.float32(self.lib.tnso_dist_sq(_p(p, _f32p), _p(q, _f32p), mode))

    def remap_csr(self, new_to_old_i, new_to_old_j, offsets_p, indices_p):
        """CSR over a permuted copy (query set permuted by new_to_old_i, candidate set by new_to_old_j)
        -> CSR in original index space, lists ascending."""
        n = len(new_to_old_i)
        a = np.ascontiguousarray(new_to_old_i, np.int32)
        b = np.ascontiguousarray(new_to_old_j, np.int32)
        offsets_p = np.ascontiguousarray(offsets_p, np.int64)
        indices_p = np.ascontiguousarray(indices_p, np.int32)
        offs = np.zeros(n + 1, np.int64)
        idx = np.zeros(max(len(indices_p), 1), np.int32)
        if len(indices_p) == 0:
            indices_p = np.zeros(1, np.int32)
        if len(b) == 0:
            b = np.zeros(1, np.int32)
        self.lib.tnso_remap_csr(n, _p(a, _i32p), _p(b, _i32p), _p(offsets_p, _i64p), _p(indices_p, _i32p),
                                _p(offs, _i64p), _p(idx, _i32p))
        return offs, idx[:offs[-1]]

    def num_threads(self):
        return self.lib.tnso_num_threads()


# ------------------------------------------------------------------------------------------------
# the real reference
# ------------------------------------------------------------------------------------------------
def _load_ref(strict: bool):
    path = REF_STRICT_SO if strict else REF_SO
    if not os.path.exists(path):
        raise FileNotFoundError(path + " (run `make -C oracle ref` where /root/reference exists)")
    L = C.CDLL(path)
    L.ref_tns_create.restype = C.c_void_p
    L.ref_bf_create.restype = C.c_void_p
    L.ref_tns_get_counts.restype = C.c_int64
    L.ref_bf_get_counts.restype = C.c_int64
    vp = C.c_void_p
    L.ref_tns_destroy.argtypes = [vp]
    L.ref_tns_add_point_set_f.argtypes = [vp, _f32p, _f32p, C.c_int]
    L.ref_tns_add_point_set_d.argtypes = [vp, _f64p, _f64p, C.c_int]
    L.ref_tns_resize_point_set_f.argtypes = [vp, C.c_int, _f32p, _f32p, C.c_int]
    L.ref_tns_resize_point_set_d.argtypes = [vp, C.c_int, _f64p, _f64p, C.c_int]
    L.ref_tns_set_search_radius.argtypes = [vp, C.c_float]
    L.ref_tns_set_cell_size.argtypes = [vp, C.c_float]
    for name in ("ref_tns_set_symmetric_search", "ref_tns_set_all_searches", "ref_tns_set_n_threads",
                 "ref_tns_set_recursion_cap"):
        getattr(L, name).argtypes = [vp, C.c_int]
    L.ref_tns_set_active_search.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.ref_tns_set_active_search_all.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    for name in ("ref_tns_run", "ref_tns_run_scalar", "ref_tns_prepare_zsort", "ref_tns_get_n_sets"):
        getattr(L, name).argtypes = [vp]
    L.ref_tns_get_n_points_in_set.argtypes = [vp, C.c_int]
    L.ref_tns_is_search_active.argtypes = [vp, C.c_int, C.c_int]
    L.ref_tns_get_zsort_order.argtypes = [vp, C.c_int, _i32p]
    L.ref_tns_get_world_box.argtypes = [vp, _f32p]
    L.ref_tns_get_cell_size.argtypes = [vp]
    L.ref_tns_get_cell_size.restype = C.c_float
    L.ref_tns_apply_zsort_f.argtypes = [vp, C.c_int, _f32p, C.c_int]
    L.ref_tns_apply_zsort_i.argtypes = [vp, C.c_int, _i32p, C.c_int]
    L.ref_tns_apply_zsort_d.argtypes = [vp, C.c_int, _f64p, C.c_int]
    L.ref_tns_get_counts.argtypes = [vp, C.c_int, C.c_int, _i32p]
    L.ref_tns_get_lists.argtypes = [vp, C.c_int, C.c_int, _i64p, _i32p, C.c_int]
    L.ref_bf_destroy.argtypes = [vp]
    L.ref_bf_add_point_set.argtypes = [vp, _f32p, _f32p, C.c_int]
    L.ref_bf_add_point_set_r.argtypes = [vp, _f32p, C.c_float, C.c_int]
    L.ref_bf_resize_point_set.argtypes = [vp, C.c_int, _f32p, _f32p, C.c_int]
    L.ref_bf_resize_point_set_r.argtypes = [vp, C.c_int, _f32p, C.c_float, C.c_int]
    L.ref_bf_set_active_search.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    for name in ("ref_bf_set_all_searches", "ref_bf_set_symmetric_search", "ref_bf_set_n_threads"):
        getattr(L, name).argtypes = [vp, C.c_int]
    L.ref_bf_run.argtypes = [vp]
    L.ref_bf_get_counts.argtypes = [vp, C.c_int, C.c_int, _i32p]
    L.ref_bf_get_lists.argtypes = [vp, C.c_int, C.c_int, _i64p, _i32p]
    L.ref_bf_compare.argtypes = [vp, vp]
    return L


_REF_LIBS = {}


def ref_lib(strict: bool):
    if strict not in _REF_LIBS:
        _REF_LIBS[strict] = _load_ref(strict)
    return _REF_LIBS[strict]


class RefTreeNSearch:
    """The reference's tns::TreeNSearch (TreeNSearch.h:28-427) through oracle/ref_wrap.cpp.
    Arrays handed in are kept alive by this object (the reference stores raw pointers)."""

    def __init__(self, strict: bool = False):
        self.L = ref_lib(strict)
        self.h = C.c_void_p(self.L.ref_tns_create())
        self._keep = {}

    def __del__(self):
        try:
            self.L.ref_tns_destroy(self.h)
        except Exception:
            pass

    def _hold(self, key, *arrs):
        self._keep[key] = arrs

    def add_point_set(self, points, radii=None):
        dbl = points.dtype == np.float64
        points = np.ascontiguousarray(points)
        if radii is not None:
            radii = np.ascontiguousarray(radii, dtype=points.dtype)
        n = points.size // 3
        if dbl:
            s = self.L.ref_tns_add_point_set_d(self.h, _p(points, _f64p), _p(radii, _f64p), n)
        else:
            assert points.dtype == np.float32
            s = self.L.ref_tns_add_point_set_f(self.h, _p(points, _f32p), _p(radii, _f32p), n)
        self._hold(s, points, radii)
        return s

    def resize_point_set(self, s, points, radii=None, n=None):
        dbl = points.dtype == np.float64
        n = points.size // 3 if n is None else n
        if dbl:
            self.L.ref_tns_resize_point_set_d(self.h, s, _p(points, _f64p), _p(radii, _f64p), n)
        else:
            self.L.ref_tns_resize_point_set_f(self.h, s, _p(points, _f32p), _p(radii, _f32p), n)
        self._hold(s, points, radii)

    def set_search_radius(self, r): self.L.ref_tns_set_search_radius(self.h, C.c_float(float(r)))
    def set_cell_size(self, c): self.L.ref_tns_set_cell_size(self.h, C.c_float(float(c)))
    def set_symmetric_search(self, on): self.L.ref_tns_set_symmetric_search(self.h, int(on))
    def set_all_searches(self, on): self.L.ref_tns_set_all_searches(self.h, int(on))
    def set_n_threads(self, n): self.L.ref_tns_set_n_threads(self.h, int(n))
    def set_recursion_cap(self, n): self.L.ref_tns_set_recursion_cap(self.h, int(n))

    def set_active_search(self, i, j, active=True):
        self.L.ref_tns_set_active_search(self.h, int(i), int(j), int(active))

    def run(self): self.L.ref_tns_run(self.h)
    def run_scalar(self): self.L.ref_tns_run_scalar(self.h)
    def prepare_zsort(self): self.L.ref_tns_prepare_zsort(self.h)
    def get_n_points_in_set(self, s): return self.L.ref_tns_get_n_points_in_set(self.h, s)

    def get_world_box(self):
        """{bottom[3], top[3]} of the reference's private world box (TreeNSearch.h:400)."""
        out = np.zeros(6, np.float32)
        self.L.ref_tns_get_world_box(self.h, _p(out, _f32p))
        return out

    def get_cell_size(self):
        return np.float32(self.L.ref_tns_get_cell_size(self.h))

    def get_zsort_order(self, s):
        out = np.zeros(max(self.get_n_points_in_set(s), 1), np.int32)
        self.L.ref_tns_get_zsort_order(self.h, s, _p(out, _i32p))
        return out[:self.get_n_points_in_set(s)]

    def apply_zsort(self, s, data, stride=1):
        fn = {np.dtype(np.float32): (self.L.ref_tns_apply_zsort_f, _f32p),
              np.dtype(np.int32): (self.L.ref_tns_apply_zsort_i, _i32p),
              np.dtype(np.float64): (self.L.ref_tns_apply_zsort_d, _f64p)}[data.dtype]
        fn[0](self.h, s, _p(data, fn[1]), stride)

    def neighbor_csr(self, i, j, sort_each=True):
        """(offsets int64[n+1], indices int32[total]) of pair (i,j)."""
        n = self.get_n_points_in_set(i)
        counts = np.zeros(max(n, 1), np.int32)
        total = self.L.ref_tns_get_counts(self.h, i, j, _p(counts, _i32p))
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(counts[:n], out=offsets[1:])
        indices = np.zeros(max(total, 1), np.int32)
        self.L.ref_tns_get_lists(self.h, i, j, _p(offsets, _i64p), _p(indices, _i32p), int(sort_each))
        return offsets, indices[:total]


class RefBruteforce:
    """The reference's tests/BruteforceNSearch (BruteforceNSearch.h:17-51)."""

    def __init__(self, strict: bool = False):
        self.L = ref_lib(strict)
        self.h = C.c_void_p(self.L.ref_bf_create())
        self._keep = {}
        self._n = {}

    def __del__(self):
        try:
            self.L.ref_bf_destroy(self.h)
        except Exception:
            pass

    def add_point_set(self, points, radii):
        points = np.ascontiguousarray(points, np.float32)
        n = points.size // 3
        if np.isscalar(radii) or getattr(radii, "ndim", 1) == 0:
            s = self.L.ref_bf_add_point_set_r(self.h, _p(points, _f32p), C.c_float(float(radii)), n)
        else:
            radii = np.ascontiguousarray(radii, np.float32)
            s = self.L.ref_bf_add_point_set(self.h, _p(points, _f32p), _p(radii, _f32p), n)
        self._keep[s] = (points, radii)
        self._n[s] = n
        return s

    def resize_point_set(self, s, points, radii, n=None):
        n = points.size // 3 if n is None else n
        if np.isscalar(radii) or getattr(radii, "ndim", 1) == 0:
            self.L.ref_bf_resize_point_set_r(self.h, s, _p(points, _f32p), C.c_float(float(radii)), n)
        else:
            self.L.ref_bf_resize_point_set(self.h, s, _p(points, _f32p), _p(radii, _f32p), n)
        self._keep[s] = (points, radii)
        self._n[s] = n

    def set_active_search(self, i, j, active=True): self.L.ref_bf_set_active_search(self.h, i, j, int(active))
    def set_all_searches(self, on): self.L.ref_bf_set_all_searches(self.h, int(on))
    def set_symmetric_search(self, on): self.L.ref_bf_set_symmetric_search(self.h, int(on))
    def run(self): self.L.ref_bf_run(self.h)

    def neighbor_csr(self, i, j):
        n = self._n[i]
        counts = np.zeros(max(n, 1), np.int32)
        total = self.L.ref_bf_get_counts(self.h, i, j, _p(counts, _i32p))
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(counts[:n], out=offsets[1:])
        indices = np.zeros(max(total, 1), np.int32)
        self.L.ref_bf_get_lists(self.h, i, j, _p(offsets, _i64p), _p(indices, _i32p))
        return offsets, indices[:total]

    def compare(self, tns: RefTreeNSearch) -> bool:
        return bool(self.L.ref_bf_compare(self.h, tns.h))
