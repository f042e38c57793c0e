"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports
every symbol include/pogs_amd.h declares; enum layouts match the reference's
(src/interface_c/pogs_c.h:51-69, pinned by tests/test_c_interface.cpp:149-154)."""
import ctypes
import os
import re

import pogs_amd
from pogs_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = open(os.path.join(ROOT, "include", "pogs_amd.h")).read()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    names = re.findall(r"\b(?:int|void|const char \*)\s*(Pogs\w*)\s*\(", body)
    return sorted(set(names))


def test_library_loads_and_exports_every_declared_symbol():
    names = declared_functions()
    assert set(names) == set(_lib.ABI_SYMBOLS), set(names) ^ set(_lib.ABI_SYMBOLS)
    for n in names:
        assert getattr(_lib.lib, n) is not None
    for n in ("PogsD", "PogsS", "PogsSparseD", "PogsSparseS"):  # the reference's graph-form symbols
        assert n in names


def _enum(name):
    m = re.search(r"enum\s+%s\s*\{(.*?)\}" % name, HEADER, flags=re.S)
    items = [t.strip() for t in re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S).split(",") if t.strip()]
    out, nxt = {}, 0
    for it in items:
        if "=" in it:
            k, v = [s.strip() for s in it.split("=")]
            nxt = int(v)
        else:
            k = it
        out[k] = nxt
        nxt += 1
    return out


def test_enum_layout_matches_reference():
    fn = _enum("FUNCTION")
    assert fn["ABS"] == 0 and fn["SQUARE"] == 14 and fn["ZERO"] == 15  # tests/test_c_interface.cpp:149-154
    order = ["ABS", "EXP", "HUBER", "IDENTITY", "INDBOX01", "INDEQ0", "INDGE0", "INDLE0", "LOGISTIC", "MAXNEG0",
             "MAXPOS0", "NEGENTR", "NEGLOG", "RECIPR", "SQUARE", "ZERO"]
    assert [fn[k] for k in order] == list(range(16))
    assert _enum("ORD") == {"COL_MAJ": 0, "ROW_MAJ": 1}
    st = _enum("POGS_STATUS")
    assert st["POGS_SUCCESS"] == 0 and st["POGS_MAX_ITER"] == 3 and st["POGS_NAN_FOUND"] == 4 and st["POGS_ERROR"] == 6
    # the Python mirror uses the same numbers (python/pogs/graph.py:114-132)
    for k, v in pogs_amd.Function.__members__.items():
        assert fn[k[1:].upper()] == int(v)
    assert int(pogs_amd.Ordering.ROW_MAJ) == 1


def test_struct_sizes_are_stable():
    assert ctypes.sizeof(_lib.PogsAmdDist) == 4 + 4 + 8 + 128
    assert ctypes.sizeof(_lib.PogsAmdOptions) == 32
    assert ctypes.sizeof(_lib.PogsAmdStats) % 8 == 0


def test_every_part1_entry_cites_the_reference_interface():
    for n in ("PogsD", "PogsS", "PogsSparseD", "PogsSparseS"):
        i = HEADER.index("int %s(" % n)
        assert "replaces: src/interface_c/pogs_c.h" in HEADER[max(0, i - 400):i]


def test_one_dense_solver_factory_per_streaming_shape_and_type():
    """csrc/stream.h lists the streaming shapes (POGS_STREAM_PLANS); pogs_amd/build.py compiles
    dense_plan.hip once per shape and arithmetic type, plus the windowed form, and abi.hip picks
    the factory from the shape.  Every one of them has to be in the library."""
    import re
    import subprocess

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    text = open(os.path.join(root, "pogs_amd", "csrc", "stream.h")).read()
    body = text[text.index("#define POGS_STREAM_PLANS(X)"):]
    body = body[:body.index("\n//")]
    shapes = re.findall(r"X\((\d+),\s*(\d+)\)", body)
    assert len(shapes) >= 10
    lib = os.path.join(root, "pogs_amd", "libpogs_amd.so")
    syms = subprocess.run(["nm", "-C", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    for t in ("f32", "f64"):
        for tpb, nv in shapes:
            assert "make_dense_solver_%s_p%s_%s(" % (t, tpb, nv) in syms, (t, tpb, nv)
        assert "make_dense_solver_%s_xl(" % t in syms, t


C_DRIVER = os.path.join(ROOT, "tests", "c_caller", "graph_form_driver.c")


def build_c_driver(out_dir):
    """gcc -std=c99 build of tests/c_caller/graph_form_driver.c against libpogs_amd.so -> path."""
    import subprocess

    exe = os.path.join(str(out_dir), "graph_form_driver")
    lib_dir = os.path.join(ROOT, "pogs_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-I", os.path.join(ROOT, "include"),
                        C_DRIVER, "-o", exe, "-L", lib_dir, "-lpogs_amd", "-Wl,-rpath," + lib_dir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_a_c_caller_links(tmp_path):
    """The boundary is a C ABI: include/pogs_amd.h compiles as strict C99 and as C++, and a C program
    calling PogsD / PogsS / PogsSparseD the way examples/c/lasso.c:102-106 does links against
    libpogs_amd.so with nothing but -lpogs_amd (no HIP, no C++ runtime on the caller's side).
    (tests/test_gpu_boundary.py runs it.)"""
    import subprocess

    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c"],
                ["g++", "-std=c++11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++"]):
        r = subprocess.run(cmd + ["-I", inc, "-"], input='#include "pogs_amd.h"\nint main(void) { return (int)sizeof(PogsAmdStats) == 0; }\n',
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    exe = build_c_driver(tmp_path)
    r = subprocess.run(["ldd", exe], capture_output=True, text=True)
    assert "libpogs_amd.so" in r.stdout
    assert subprocess.run([exe], capture_output=True).returncode == 2      # usage error, before any GPU call


def test_the_build_leaves_a_libpogs_cpu_alias_the_reference_loader_finds():
    """python/pogs/graph.py:29-67 looks for `libpogs_cpu.so` in its package directory: the build leaves
    that name next to libpogs_amd.so (pogs_amd/build.py), and what it points to exports the four
    graph-form entry points the reference's bindings declare (graph.py:167-233)."""
    alias = os.path.join(ROOT, "pogs_amd", "libpogs_cpu.so")
    assert os.path.exists(alias), "pogs_amd/build.py did not create the alias"
    assert os.path.samefile(alias, os.path.join(ROOT, "pogs_amd", "libpogs_amd.so"))
    lib = ctypes.CDLL(alias)
    for sym in ("PogsD", "PogsS", "PogsSparseD", "PogsSparseS"):
        getattr(lib, sym)


def test_a_null_coefficient_array_is_refused_by_the_array_entry_points(capfd):
    """The reference's four entry points take twelve coefficient arrays (src/interface_c/pogs_c.h:75-119); a NULL
    among them is a caller's mistake.  Since the *Fn entry points read a null field as a broadcast scalar, the
    array entry points must say no themselves -- before a solver is built, so this runs without a GPU: the call
    returns POGS_ERROR (6, pogs_c.h:20-26 via PogsStatus) and names the vector."""
    import numpy as np

    m, n = 6, 4
    A = np.ones((m, n), np.float32)
    one_m, one_n = np.ones(m, np.float32), np.ones(n, np.float32)
    hm, hn = np.zeros(m, np.int32), np.zeros(n, np.int32)
    x, y, l_ = np.zeros(n, np.float32), np.zeros(m, np.float32), np.zeros(m, np.float32)
    opt, it = ctypes.c_float(0), ctypes.c_uint(0)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    f = [P(one_m)] * 5 + [P(hm)]
    g = [P(one_n)] * 5 + [P(hn)]
    for which, pos in (("f", 2), ("g", 5), ("g", 0)):
        ff, gg = list(f), list(g)
        (ff if which == "f" else gg)[pos] = None
        rc = _lib.lib.PogsS(_lib.ROW_MAJ, m, n, P(A), *ff, *gg, 1.0, 1e-4, 1e-4, 10, 0, 1, 1, P(x), P(y), P(l_),
                            ctypes.byref(opt), ctypes.byref(it))
        assert rc == 6, rc
        msg = _lib.last_error()
        assert "null coefficient array" in msg and "description of %s" % which in msg, msg
    capfd.readouterr()
