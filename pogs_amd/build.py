"""Builds pogs_amd/libpogs_amd.so from pogs_amd/csrc/*.hip for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot (it is git-ignored,
not gpurun-ignored).  Usage: python pogs_amd/build.py [--force]
"""
import concurrent.futures
import os
import re
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(_HERE, "libpogs_amd.so")
# The reference's Python layer looks for `libpogs_cpu.so` in its package directory
# (python/pogs/graph.py:29-67).  The build leaves an alias of that name next to the library: copied (or
# linked) into the reference's `pogs/` directory it is picked up by the UNMODIFIED loader
# (INTEGRATION.md section 1).
ALIAS = os.path.join(_HERE, "libpogs_cpu.so")
# The test-suite's communicators (ranks as threads / as processes on ONE GPU) are a transport plug-in of the
# library (csrc/transport_plugin.h), built next to its source under tests/ -- test infrastructure, not product.
PLUGIN_SRC = os.path.join(_HERE, "..", "tests", "transport", "test_transport.hip")
PLUGIN_LIB = os.path.join(_HERE, "..", "tests", "transport", "libpogs_test_transport.so")
SOURCES = ["abi.hip", "sparse.hip", "gemm.hip", "vec_kernels.hip", "dist.hip"]
# dense_plan.hip is compiled once per arithmetic type and streaming shape (csrc/stream.h:
# POGS_STREAM_PLANS) plus the windowed form -- one small code object per shape, see dense_plan.hip
PLAN_SOURCE = "dense_plan.hip"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# Floating-point contraction OFF everywhere but gemm.hip: a product is fused with a sum exactly where the source says
# fma (csrc/stream.h: vdot / vfma / vscale_add, csrc/sell.h: the SpMV's accumulate) and nowhere else, so the roundings
# of the iteration no longer follow instruction selection (round 5: an edit to a reduction changed which products of the
# row dots were fused in OTHER kernels and moved a 33 x 40001 problem by 1e-4).  The scalar algebra around the dots --
# prox, over-relaxation, residual sums -- then runs unfused, which is also what the reference's x86-64 build does.
# gemm.hip keeps the default: its products are the matrix cores' (fixed by the hardware) and the scalar loops of the
# diagonal-block Cholesky, where an unfused multiply-subtract would double the single-workgroup chain.
NO_CONTRACT = "-ffp-contract=off"
CONTRACT_DEFAULT_SOURCES = ("gemm.hip",)
# development aid: extra -D flags for EVERY object (the experiment switches that sit in inline functions shared by
# all translation units, e.g. POGS_C3_LEAN_BLOCKS; use a separate checkout: the objects are not tracked per flag set)
FLAGS += os.environ.get("POGS_AMD_EXTRA_FLAGS", "").split()


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newest_dep():
    t = 0.0
    for root in (CSRC, os.path.join(_HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip")):
                t = max(t, os.path.getmtime(os.path.join(root, f)))
    return t


def _plan_jobs():
    """(object name, extra -D flags) for every build of dense_plan.hip."""
    text = open(os.path.join(CSRC, "stream.h")).read()
    body = text[text.index("#define POGS_STREAM_PLANS(X)"):]
    body = body[:body.index("\n//")]
    shapes = [(int(a), int(b)) for a, b in re.findall(r"X\((\d+),\s*(\d+)\)", body)]
    if not shapes:
        raise RuntimeError("POGS_STREAM_PLANS not found in stream.h")
    jobs = []
    for tname, ctype in (("f32", "float"), ("f64", "double")):
        for tpb, nv in shapes:
            name = "%s_p%d_%d" % (tname, tpb, nv)
            jobs.append(("dense_%s.o" % name, ["-DPOGS_PLAN_T=%s" % ctype, "-DPOGS_PLAN_NAME=%s" % name,
                                                "-DPOGS_PLAN_TPB=%d" % tpb, "-DPOGS_PLAN_NV=%d" % nv]))
        name = "%s_xl" % tname
        jobs.append(("dense_%s.o" % name, ["-DPOGS_PLAN_T=%s" % ctype, "-DPOGS_PLAN_NAME=%s" % name, "-DPOGS_PLAN_XL=1"]))
    return jobs


def _deps_newer_than(obj):
    """True if any file the object was compiled from (its -MMD dependency list) is newer than it, or the
    list is missing: a header change rebuilds the objects that include it, not all 37."""
    dep = obj[:-2] + ".d"
    if not os.path.exists(dep):
        return True
    t_obj = os.path.getmtime(obj)
    text = open(dep).read().replace("\\\n", " ")
    files = text.split(":", 1)[1].split() if ":" in text else []
    for f in files:
        if not os.path.exists(f) or os.path.getmtime(f) > t_obj:
            return True
    return False


def _compile(job):
    src, objname, defs = job
    obj = os.path.join(OBJ, objname)
    # (an explicit -ffp-contract in POGS_AMD_EXTRA_FLAGS -- an A / B build -- takes precedence)
    fp = [] if src in CONTRACT_DEFAULT_SOURCES or any(f.startswith("-ffp-contract") for f in FLAGS) else [NO_CONTRACT]
    cmd = [_hipcc()] + FLAGS + fp + defs + ["-MMD", "-MF", obj[:-2] + ".d", "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s (%s):\n%s\n%s" % (src, objname, r.stdout, r.stderr))
    return obj


def _alias():
    try:
        if os.path.islink(ALIAS) or os.path.exists(ALIAS):
            if os.path.islink(ALIAS) and os.readlink(ALIAS) == os.path.basename(LIB):
                return
            os.remove(ALIAS)
        os.symlink(os.path.basename(LIB), ALIAS)
    except OSError:
        pass   # a file system without symlinks: the alias is a convenience, not a dependency


def build_plugin(force=False, verbose=False):
    """tests/transport/libpogs_test_transport.so (skipped when the tests directory is not there)."""
    if not os.path.exists(PLUGIN_SRC):
        return None
    hdr = os.path.join(CSRC, "transport_plugin.h")
    if not force and os.path.exists(PLUGIN_LIB) and \
            os.path.getmtime(PLUGIN_LIB) >= max(os.path.getmtime(PLUGIN_SRC), os.path.getmtime(hdr)):
        return PLUGIN_LIB
    if verbose:
        print("pogs_amd.build: compiling the test transport plug-in", file=sys.stderr)
    cmd = [_hipcc()] + FLAGS + [NO_CONTRACT, "-shared", PLUGIN_SRC, "-o", PLUGIN_LIB, "-lrt"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for the test transport plug-in:\n%s\n%s" % (r.stdout, r.stderr))
    return PLUGIN_LIB


def build(force=False, verbose=False):
    """Compile (if stale) and return the path of libpogs_amd.so."""
    build_plugin(force, verbose)
    dep = _newest_dep()
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= dep:
        _alias()
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    todo = []
    objs = []
    jobs = [(src, src.replace(".hip", ".o"), []) for src in SOURCES]
    jobs += [(PLAN_SOURCE, objname, defs) for objname, defs in _plan_jobs()]
    for job in jobs:
        src, objname, _ = job
        obj = os.path.join(OBJ, objname)
        objs.append(obj)
        stale = force or not os.path.exists(obj) or _deps_newer_than(obj)
        if stale:
            todo.append(job)
    if verbose:
        print("pogs_amd.build: compiling", [j[1] for j in todo], file=sys.stderr)
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(_compile, todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    _alias()
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
