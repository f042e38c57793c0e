"""Why does the compiled reference (oracle/_ref/libpogs_cpu.so) slow down far more than linearly
beyond ~40000 rows on the GPU box?  ("next round" item 1a.)

Prints the container's limits (cgroup cpu / memory, /dev/shm, THP, NUMA), then runs the reference
on the leading `rows` rows of one C2-style matrix under several environments while a monitor
samples cgroup cpu.stat (throttling), memory.current / memory.stat (reclaim, major faults) and the
child's RSS / thread count / CPU seconds; the child's stdout lines are time-stamped so that setup
and loop can be told apart.

    python scripts/ref_collapse_probe.py "40000,70000,100000" 10000 240 [variant,variant...]
"""
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def rd(path, default=""):
    try:
        return open(path).read().strip()
    except Exception:
        return default


def kv(path):
    out = {}
    for ln in rd(path).splitlines():
        p = ln.split()
        if len(p) == 2:
            try:
                out[p[0]] = int(p[1])
            except ValueError:
                pass
    return out


def limits():
    info = {
        "nproc_visible": os.cpu_count(),
        "sched_affinity": len(os.sched_getaffinity(0)),
        "cpu.max": rd("/sys/fs/cgroup/cpu.max"),
        "memory.max": rd("/sys/fs/cgroup/memory.max"),
        "memory.high": rd("/sys/fs/cgroup/memory.high"),
        "memory.swap.max": rd("/sys/fs/cgroup/memory.swap.max"),
        "memory.current": rd("/sys/fs/cgroup/memory.current"),
        "cpuset.cpus.effective": rd("/sys/fs/cgroup/cpuset.cpus.effective"),
        "cpuset.mems.effective": rd("/sys/fs/cgroup/cpuset.mems.effective"),
        "thp_enabled": rd("/sys/kernel/mm/transparent_hugepage/enabled"),
        "thp_defrag": rd("/sys/kernel/mm/transparent_hugepage/defrag"),
        "numa_nodes": len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")])
        if os.path.isdir("/sys/devices/system/node") else None,
        "meminfo": {k: v for k, v in (ln.split(":") for ln in rd("/proc/meminfo").splitlines()[:6])},
    }
    try:
        info["df_shm"] = subprocess.run(["df", "-h", "/dev/shm", "/tmp"], capture_output=True, text=True).stdout
    except Exception:
        pass
    try:
        info["lscpu"] = [ln for ln in subprocess.run(["lscpu"], capture_output=True, text=True).stdout.splitlines()
                         if any(k in ln for k in ("Model name", "Socket", "NUMA", "Thread(s)", "Core(s)", "L3"))]
    except Exception:
        pass
    return info


class Monitor(threading.Thread):
    def __init__(self, pid, every=2.0):
        super().__init__(daemon=True)
        self.pid, self.every, self.rows, self.stop = pid, every, [], False

    def run(self):
        t0 = time.time()
        while not self.stop:
            cs = kv("/sys/fs/cgroup/cpu.stat")
            ms = kv("/sys/fs/cgroup/memory.stat")
            st = rd("/proc/%d/status" % self.pid)
            rss = thr = 0
            for ln in st.splitlines():
                if ln.startswith("VmRSS"):
                    rss = int(ln.split()[1]) // 1024
                if ln.startswith("Threads"):
                    thr = int(ln.split()[1])
            stat = rd("/proc/%d/stat" % self.pid).split()
            ut = (int(stat[13]) + int(stat[14])) / os.sysconf("SC_CLK_TCK") if len(stat) > 15 else 0.0
            self.rows.append({"t": round(time.time() - t0, 1), "rss_mb": rss, "threads": thr, "cpu_s": round(ut, 1),
                              "throttled_ms": cs.get("throttled_usec", 0) // 1000, "nr_throttled": cs.get("nr_throttled", 0),
                              "mem_cur_mb": int(rd("/sys/fs/cgroup/memory.current", "0") or 0) >> 20,
                              "pgmajfault": ms.get("pgmajfault", 0), "pgscan": ms.get("pgscan", 0),
                              "shmem_mb": ms.get("shmem", 0) >> 20, "file_mb": ms.get("file", 0) >> 20,
                              "anon_mb": ms.get("anon", 0) >> 20, "thp_mb": ms.get("anon_thp", 0) >> 20})
            time.sleep(self.every)


NT = os.environ.get("PROBE_THREADS", "16")
VARIANTS = {
    # name: (env overrides, input dir kind, mmap input)
    "shm16": ({"MKL_NUM_THREADS": NT, "OMP_NUM_THREADS": NT, "MKL_DYNAMIC": "FALSE"}, "shm", False),
    "disk16": ({"MKL_NUM_THREADS": NT, "OMP_NUM_THREADS": NT, "MKL_DYNAMIC": "FALSE"}, "disk", False),
    "gnu16": ({"MKL_NUM_THREADS": NT, "OMP_NUM_THREADS": NT, "MKL_DYNAMIC": "FALSE", "MKL_THREADING_LAYER": "GNU"},
              "disk", False),
    "seq": ({"MKL_NUM_THREADS": "1", "OMP_NUM_THREADS": "1", "MKL_THREADING_LAYER": "SEQUENTIAL"}, "disk", False),
    "aff16": ({"MKL_NUM_THREADS": NT, "OMP_NUM_THREADS": NT, "MKL_DYNAMIC": "FALSE", "PROBE_TASKSET": "16"},
              "disk", False),
    "passive16": ({"MKL_NUM_THREADS": NT, "OMP_NUM_THREADS": NT, "MKL_DYNAMIC": "FALSE",
                   "OMP_WAIT_POLICY": "PASSIVE", "KMP_BLOCKTIME": "0"}, "disk", False),
    "gnupassive16": ({"MKL_NUM_THREADS": NT, "OMP_NUM_THREADS": NT, "MKL_DYNAMIC": "FALSE", "MKL_THREADING_LAYER": "GNU",
                      "OMP_WAIT_POLICY": "PASSIVE"}, "shm", False),
    "t8": ({"MKL_NUM_THREADS": "8", "OMP_NUM_THREADS": "8", "MKL_DYNAMIC": "FALSE", "KMP_BLOCKTIME": "0"}, "disk", False),
}


def run_variant(name, A, f, g, timeout):
    import oracle_binding as ob

    env_over, where, _ = VARIANTS[name]
    base = "/dev/shm" if where == "shm" else os.path.join(ROOT, "gpurun_out")
    os.makedirs(base, exist_ok=True)
    td = tempfile.mkdtemp(dir=base)
    t_w = time.time()
    np.save(os.path.join(td, "A.npy"), A)
    payload = {"dtype": "float32", "params": np.array([1.0, 1e-4, 1e-4, 2500, 2, 1, 1, 1], dtype=np.float64)}
    for k in "habcde":
        payload["f_" + k] = np.asarray(f[k])
        payload["g_" + k] = np.asarray(g[k])
    np.savez(os.path.join(td, "in.npz"), **payload)
    t_w = time.time() - t_w
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    taskset = env_over.get("PROBE_TASKSET")
    env.update({k: v for k, v in env_over.items() if k != "PROBE_TASKSET"})
    cmd = [sys.executable, "-u", os.path.join(ROOT, "tests", "ref_runner.py"), td]
    if taskset:
        cpus = sorted(os.sched_getaffinity(0))[:int(taskset)]
        cmd = ["taskset", "-c", ",".join(str(c) for c in cpus)] + cmd
    t0 = time.time()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    mon = Monitor(proc.pid)
    mon.start()
    stamps = []

    def reader():
        for ln in proc.stdout:
            stamps.append((round(time.time() - t0, 2), ln.rstrip()[:110]))

    rt = threading.Thread(target=reader, daemon=True)
    rt.start()
    timed_out = False
    try:
        proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        proc.kill()
        proc.wait()
    wall = time.time() - t0
    mon.stop = True
    rt.join(2)
    res = {"variant": name, "rows": A.shape[0], "write_s": round(t_w, 1), "wall_s": round(wall, 1), "timed_out": timed_out}
    iters = [s for s in stamps if s[1].strip()[:1].isdigit()]
    if iters:
        res["first_iter_line_at_s"] = iters[0][0]
        res["last_iter_line"] = iters[-1]
        res["iter_lines"] = len(iters)
    for s in stamps:
        if "Total" in s[1] or "Status" in s[1]:
            res.setdefault("summary", []).append(s)
    # monitor: first, a few in the middle, last
    rows = mon.rows
    pick = rows[::max(1, len(rows) // 8)] + rows[-1:]
    res["monitor"] = pick
    subprocess.call(["rm", "-rf", td])
    return res


def main():
    from pogs_amd import graph as G
    from pogs_amd import synth

    rows_list = [int(v) for v in sys.argv[1].split(",")]
    n = int(sys.argv[2])
    timeout = float(sys.argv[3])
    variants = sys.argv[4].split(",") if len(sys.argv) > 4 else ["shm16", "disk16"]
    print(json.dumps({"limits": limits()}, indent=1), flush=True)
    m = max(rows_list)
    t0 = time.time()
    A, b, _ = synth.dense_lasso_rows(m, n, seed=2024)
    print("generated %dx%d in %.1f s" % (m, n, time.time() - t0), flush=True)
    for rows in rows_list:
        f, g = G.lasso_functions(b[:rows], 0.1, n)
        fs = {k: getattr(f, k) for k in "habcde"}
        gs = {k: getattr(g, k) for k in "habcde"}
        for v in variants:
            r = run_variant(v, A[:rows], fs, gs, timeout)
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
