"""CPU tests: liblpx.so loads without a GPU, exports every symbol include/lpx.h declares, and every
compute entry point fails loudly (no CPU fallback) when no device is visible."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):          # lpx.h (the boundary) and lpx_test.h (test-only entry)
        if not h.endswith(".h"):
            continue
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        names |= set(re.findall(r"\b(lpx_[a-z0-9_]+)\s*\(", hdr))
    names -= {"lpx_pivot_cb"}
    return sorted(names)


def test_boundary_header_carries_no_test_hooks():
    """The struct every host must mirror (lpx_solve_opts) has no test seam in it; the stand-ins of the CPU test-suite
    live behind include/lpx_test.h and nothing in the package installs them except when a test asks for it."""
    hdr = open(os.path.join(ROOT, "include", "lpx.h")).read()
    assert "test_node_lp" not in hdr and "test_knap_relax" not in hdr and "lpx_test_set_seams" not in hdr
    assert "lpx_test_set_seams" in open(os.path.join(ROOT, "include", "lpx_test.h")).read()


def test_header_declares_something():
    names = declared_symbols()
    assert "lpx_primal_run" in names and "lpx_tableau_create" in names and len(names) >= 15


def test_every_declared_symbol_is_exported(lpx):
    L = lpx._lib.lib()
    missing = [n for n in declared_symbols() if not hasattr(L, n)]
    assert not missing, missing


def test_abi_version(lpx):
    assert lpx._lib.lib().lpx_abi_version() == 1


def test_no_cpu_fallback_without_device(lpx):
    L = lpx._lib.lib()
    if L.lpx_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(lpx.LpxError) as ei:
        lpx.DeviceTableau(4, 6)
    assert ei.value.code == lpx._lib.EDEVICE
    T = np.zeros((3, 5))
    basis = np.zeros(2, np.int32)
    with pytest.raises(lpx.LpxError):
        lpx.primal_tableau(T, basis)
    with pytest.raises(lpx.LpxError):
        lpx.dual_tableau(T, basis)


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "linear_programming_solver_lpr381_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.lower(), f"{f} mentions the oracle"


def test_only_tests_smoke_and_the_cpu_baseline_use_the_oracle():
    """tools/ never imports the oracle; bench.py does so only inside its cpu_baseline leg; __graft_entry__ only in
    build() (compiling the checker) and smoke() (checking one result)."""
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            txt = open(os.path.join(ROOT, "tools", f), errors="replace").read()
            assert "import oracle" not in txt and "from oracle" not in txt, f"tools/{f} uses the oracle"
    bench = open(os.path.join(ROOT, "bench.py")).read()
    first = bench.index("from oracle import")
    assert bench.count("from oracle import") == 1 and bench.index("CPU baseline") < first, "oracle import outside the cpu_baseline leg"
    assert "if world == 1:" in bench[bench.index("CPU baseline"):first]


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """include/lpx.h must be consumable by a C compiler (the boundary is a C ABI, not a C++ API), and a C client must
    link against liblpx.so with nothing but the header.  No GPU call is made: lpx_abi_version / lpx_format_number only."""
    import shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "lpx.h")
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    so = os.path.join(root, "linear_programming_solver_lpr381_amd", "liblpx.so")
    if not os.path.exists(so):
        pytest.skip("liblpx.so not built")
    src = tmp_path / "client.c"
    src.write_text('#include "lpx.h"\n#include <stdio.h>\n'
                   'int main(void) { char b[32]; lpx_run_opts o; lpx_default_opts(&o, 0);\n'
                   '  lpx_format_number(2.5, b, (int)sizeof b);\n'
                   '  printf("%d %s %d %g\\n", lpx_abi_version(), b, o.max_iter, o.eps); return 0; }\n')
    exe = tmp_path / "client"
    libdir = os.path.dirname(so)
    subprocess.run([gcc, "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", libdir, "-llpx",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == "1" and out[1] == "2.5" and out[2] == "10000" and float(out[3]) == 1e-9
