"""treensearch_amd -- MI355X-native fixed-radius neighbour search behind the tns::TreeNSearch API.

    csrc/         gfx950 HIP kernels + the C-ABI engine (include/tnsx.h) -> lib/libtnsx.so
    api.py        ctypes mirror of the reference class (used by tests, bench.py and Python consumers)
    datagen.py    deterministic synthetic clouds (BASELINE.json configs)
    build.py      hipcc build recipe

The C++ drop-in (`#include <TreeNSearch>`) lives in include/.  There is no CPU compute path in this package.
"""
from .api import (ARITH_CONTRACTED, ARITH_STRICT, NeighborList, TnsxError, TreeNSearch, load_library)  # noqa: F401

__all__ = ["TreeNSearch", "NeighborList", "TnsxError", "ARITH_STRICT", "ARITH_CONTRACTED", "load_library"]
