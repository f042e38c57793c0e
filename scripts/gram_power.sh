#!/bin/bash
# Power and clock telemetry of the fp16-split Gram kernel ("record power and SCLK during the kernel").
# A C2 handle is created with POGS_AMD_GRAM_REPEAT=<reps>: ~1.5 s of nothing but gram_f16s_kernel, sampled every ~50 ms with
# rocm-smi (and amd-smi where it answers); for comparison the same window over the dense pass (HBM-bound) of a C2 solve loop.
# usage: gram_power.sh <tag> [reps]      -> gpurun_out/gram_power_<tag>/{samples.jsonl,summary.json}
tag=${1:-r05}; reps=${2:-60}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gram_power_$tag
rm -rf $O; mkdir -p $O
cd $R
python - "$O" "$reps" <<'PY'
import json, os, subprocess, sys, threading, time
O, reps = sys.argv[1], int(sys.argv[2])
os.environ["POGS_AMD_TORCH_PRELOAD"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pogs_amd
from pogs_amd import synth, graph as G

stop, phase = threading.Event(), {"name": "idle"}
samples = []
def smi(cmd):
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=5).stdout
    except Exception as e:
        return "ERR %r" % (e,)
def sampler():
    while not stop.is_set():
        t = time.time()
        out = smi(["rocm-smi", "--showpower", "--showclocks", "--showuse", "--json"])
        samples.append({"t": t, "phase": phase["name"], "dt": time.time() - t, "rocm_smi": out})
        time.sleep(0.02)
probe = {"rocm_smi": smi(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showperflevel", "--json"]),
         "amd_smi": smi(["amd-smi", "metric", "-p", "-c", "--json"])[:4000]}
A = torch.randn((100000, 10000), device="cuda", dtype=torch.float32)
b = np.random.default_rng(0).standard_normal(100000)
f, g = G.lasso_functions(b, 0.1, 10000)
def create():
    return pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(100000, 10000), device_ptr=True)
create().close()   # code objects loaded, pool warm
th = threading.Thread(target=sampler); th.start()
time.sleep(0.5)
phase["name"] = "gram_repeat"
os.environ["POGS_AMD_GRAM_REPEAT"] = str(reps)
t0 = time.time(); s = create(); t_gram = time.time() - t0
os.environ.pop("POGS_AMD_GRAM_REPEAT")
phase["name"] = "idle2"; time.sleep(0.5)
phase["name"] = "dense_pass"
s.begin_run(f, g); t0 = time.time(); s.iterate(2500); t_pass = time.time() - t0
phase["name"] = "idle3"; time.sleep(0.3)
s.close()
# the native fp32 MFMA Gram for comparison (100 ms per product)
phase["name"] = "gram_fp32"
os.environ["POGS_AMD_GRAM"] = "fp32"
t0 = time.time()
for _ in range(10):
    create().close()
t_fp32 = time.time() - t0
os.environ.pop("POGS_AMD_GRAM")
phase["name"] = "end"; time.sleep(0.2)
stop.set(); th.join()
with open(os.path.join(O, "samples.jsonl"), "w") as fh:
    for smp in samples:
        fh.write(json.dumps(smp) + "\n")
def fields(txt):
    try:
        d = json.loads(txt)
    except Exception:
        return {}
    out = {}
    for card, kv in d.items():
        if not isinstance(kv, dict):
            continue
        for k, v in kv.items():
            kl = k.lower()
            try:
                if "power" in kl and "(w)" in kl: out["power_w"] = float(v)
                elif kl.startswith("sclk clock speed"): out["sclk_mhz"] = float(str(v).strip("()").lower().replace("mhz", ""))
                elif kl.startswith("mclk clock speed"): out["mclk_mhz"] = float(str(v).strip("()").lower().replace("mhz", ""))
                elif kl.startswith("gpu use"): out["gpu_use"] = float(v)
            except Exception:
                pass
        break
    return out
summary = {"probe": probe, "gram_repeat_s": t_gram, "gram_launch_ms": 1e3 * t_gram / max(reps, 1), "dense_pass_s": t_pass, "gram_fp32_10_handles_s": t_fp32,
           "sample_period_s": float(np.median([s_["dt"] for s_ in samples])) if samples else None, "phases": {}}
for ph in ("idle", "gram_repeat", "dense_pass", "gram_fp32"):
    rows = [fields(s_["rocm_smi"]) for s_ in samples if s_["phase"] == ph]
    rows = [r for r in rows if r]
    agg = {}
    for key in ("power_w", "sclk_mhz", "mclk_mhz", "gpu_use"):
        v = [r[key] for r in rows if key in r]
        if v:
            agg[key] = {"n": len(v), "min": min(v), "median": float(np.median(v)), "max": max(v)}
    summary["phases"][ph] = agg
json.dump(summary, open(os.path.join(O, "summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "probe"}, indent=1))
print("first raw sample:", samples[0]["rocm_smi"][:600] if samples else None)
PY
