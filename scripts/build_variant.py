"""Development aid: pogs_amd/variants/libpogs_amd_<tag>.so = the library with ONE source recompiled under extra -D flags
(tile-shape and kernel experiments measured side by side on one box: scripts/spmv_probe.sh).
    python scripts/build_variant.py <tag> <source.hip> [--replace=<object of the regular build>] -DNAME=VALUE ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pogs_amd", "csrc")
OBJ = os.path.join(CSRC, "build")
tag, src, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
replace = src.replace(".hip", ".o")
if defs and defs[0].startswith("--replace="):   # the object of the regular build this one stands in for (plan objects: dense_f32_p256_5.o ...)
    replace = defs[0].split("=", 1)[1]
    defs = defs[1:]
out_dir = os.path.join(ROOT, "pogs_amd", "variants")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, "%s_%s.o" % (src.replace(".hip", ""), tag))
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
if src != "gemm.hip" and not any(d.startswith("-ffp-contract") for d in defs):
    flags.append("-ffp-contract=off")   # as pogs_amd/build.py
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + defs + ["-c", os.path.join(CSRC, src), "-o", obj])
objs = [os.path.join(OBJ, f) for f in sorted(os.listdir(OBJ)) if f.endswith(".o") and f != replace] + [obj]
lib = os.path.join(out_dir, "libpogs_amd_%s.so" % tag)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
print(lib)
