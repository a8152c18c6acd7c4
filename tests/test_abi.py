"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports every symbol that
include/tnsx.h declares, and refuses to work without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tnsx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tnsx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_python_mirror_agree():
    from treensearch_amd import api
    assert _declared_symbols() == sorted(api.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(built_library):
    lib = C.CDLL(built_library)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"libtnsx.so does not export {name}"


def test_version(built_library):
    from treensearch_amd import api
    L = api.load_library()
    assert L.tnsx_version() == 600   # TNSX_VERSION of include/tnsx.h (round 6: tnsx_pair_csr_device, tnsx_get_device, tnsx_stats.nan_fixups)


def test_no_cpu_fallback(built_library):
    """Without a GPU tnsx_create must fail with TNSX_ERR_NO_DEVICE and say why."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import treensearch_amd as T
    with pytest.raises(T.TnsxError) as e:
        T.TreeNSearch()
    assert e.value.status == 2
    assert "no HIP device" in e.value.message


def test_product_never_touches_the_oracle():
    """The product path (treensearch_amd/, include/) must not import, link or mention oracle/."""
    bad = []
    for base in ("treensearch_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", "")) and not f.endswith((".so", ".o", ".pyc")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"\boracle\b", txt) and "no CPU" not in txt[:0]:
                        for line in txt.splitlines():
                            if re.search(r"(import|include|CDLL|dlopen).*oracle", line):
                                bad.append((f, line.strip()))
    assert not bad, bad


def test_no_environment_switches_in_the_engine():
    """Nothing in the engine's sources reads the environment: no run-time switch can make a timed pass skip work or change a
    launch (the store-free pool pass is a compile-time variant for tools/, the launch widths are tnsx_options fields).  The C++
    shim reads TNSX_DEVICES once, in its constructor -- the only way a drop-in user who cannot touch the call site can ask for
    the multi-device mode."""
    for dp, _, files in os.walk(os.path.join(ROOT, "treensearch_amd", "csrc")):
        for f in files:
            txt = open(os.path.join(dp, f), errors="ignore").read()
            assert "getenv" not in txt, f"{f} reads the environment"


def test_cpp_shim_compiles(built_library, tmp_path):
    """The header-only tns::TreeNSearch shim compiles with plain g++ against the C ABI (no hip headers needed)."""
    import subprocess
    src = tmp_path / "shim_smoke.cpp"
    src.write_text("""
#include <TreeNSearch>
#include <vector>
int main() {
    tns::TreeNSearch* p = nullptr; (void)p;
    std::vector<float> pts(30, 0.f);
    // compile-time check of the reference signatures (TreeNSearch.h:50-334)
    int (tns::TreeNSearch::*a)(const float*, const int) = &tns::TreeNSearch::add_point_set; (void)a;
    int (tns::TreeNSearch::*b)(const double*, const double*, const int) = &tns::TreeNSearch::add_point_set; (void)b;
    void (tns::TreeNSearch::*c)(const int, const int, const bool) = &tns::TreeNSearch::set_active_search; (void)c;
    void (tns::TreeNSearch::*d)(const int, const bool, const bool) = &tns::TreeNSearch::set_active_search; (void)d;
    tns::NeighborList (tns::TreeNSearch::*e)(const int, const int, const int) const = &tns::TreeNSearch::get_neighborlist; (void)e;
    return 0;
}
""")
    exe = tmp_path / "shim_smoke"
    lib_dir = os.path.dirname(built_library)
    subprocess.check_call(["g++", "-std=c++17", "-fopenmp", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L" + lib_dir, "-ltnsx", "-Wl,-rpath," + lib_dir])
    assert subprocess.call([str(exe)]) == 0
