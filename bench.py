"""Benchmark of the hot path: ADMM iterations/s of the graph-form solve on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4]

With N > 1 and no launcher in the environment the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`
(one rank per GPU, RCCL over xGMI); under a launcher (RANK / WORLD_SIZE set) it runs as
that rank.

Workloads (BASELINE.json `configs`); rows are PER GPU ("weak" scaling):
  c2 (default, the metric's configuration): solve_lasso, dense fp32, A = 100000 x 10000 per
      GPU, synthetic N(0,1), x_true 10% dense, b = A x_true + 0.1 N(0,1), lambda = 0.1,
      default tolerances, direct projector.  N = 8 is config C5 (800000 x 10000).
  c3: solve_logistic, dense fp32, A = 200000 x 5000, labels from logits with std 2
      (pogs_amd/synth.py: the reference recipe is nearly separable at this size), lambda = 0.01.
  c4: solve_lasso, CSR fp32, A = 2000000 x 500000 with 50 non-zeros per row, CGLS projector.

A step is ONE ADMM iteration of a real default-tolerance solve (prox, gap and tolerance
sums, over-relaxation, projection, residual bookkeeping, dual update, adaptive rho, exact
residuals whenever the reference would evaluate them); when a solve converges the next step
starts the next solve from the cold start, so K steps are K genuine iterations.  The one-time
setup (equilibration, norm estimate, Gram + Cholesky / blocked SpMV layout) is outside the
timed region (init_s); a complete cold solve is reported as time_to_converge_s.

value = N * K / T: iterations of one per-GPU shard per second, summed over ranks (at N = 1 the
ADMM it/s of the configuration).  T is the max over ranks of the time of exactly K steps
between barrier + synchronize on both sides; when K steps take less than 0.1 s several such
windows are timed back to back and the median is reported (`windows`).

Inputs are resident in HBM when the timed region starts.  The JSON line carries `roofline`
(the dominant kernel, timed with HIP events on the solver's stream over the timed region)
and, at N = 1, `cpu_baseline`: the compiled reference (oracle/_ref, clean subprocess) on the
same (A, b, lambda) on this box's host cores -- the full workload when it fits the budget --
with `parity_vs_reference` comparing the two solutions; the OpenMP oracle port for c4.
"""
import argparse
import json
import os
import socket
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 achievable by a float4 copy

CONFIGS = {
    "c2": dict(m=100000, n=10000, kind="dense_lasso", lambd=0.1, cfg_index=1),
    "c3": dict(m=200000, n=5000, kind="dense_logistic", lambd=0.01, cfg_index=2),
    "c4": dict(m=2000000, n=500000, kind="csr_lasso", lambd=0.1, nnz_per_row=50, cfg_index=3),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--m", type=int, default=0, help="rows per GPU (default: the configuration's)")
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--projector", choices=["default", "cgls"], default="default",
                    help="dense configurations: 'cgls' selects the matrix-free CGLS projector (the reference's "
                         "ProjectorCgls on a dense matrix, src/cpu/projector/projector_cgls.cpp) instead of the direct one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=150.0, help="wall-clock allowance for the CPU baseline")
    ap.add_argument("--cpu-full", action="store_true",
                    help="also run the reference on the FULL workload (minutes in the build container; the GPU box's "
                         "container -- 16-core CPU quota -- needs more than 15 minutes at C2)")
    return ap.parse_args()


def maybe_spawn(args):
    """--gpus N > 1 without a launcher: become `torch.distributed.run` with N ranks."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def make_problem(cfg, m, n, rank, dev):
    """Per-rank rows with a shared x_true / w_true (generated on the device).  Returns
    (device matrix holder, what Solver() takes, function pair builder inputs)."""
    import numpy as np
    import torch

    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    kind = cfg["kind"]
    if kind == "dense_lasso":
        x_true = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.1)
        g.manual_seed(1000 + rank)
        A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
        b = A @ x_true + 0.1 * torch.randn(m, generator=g, device=dev)
        return A, b.double().cpu().numpy()
    if kind == "dense_logistic":
        w = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.3)
        w = w * (2.0 / torch.sqrt((w * w).sum()))
        g.manual_seed(1000 + rank)
        A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
        p = torch.sigmoid(A @ w)
        lab = 2.0 * (torch.rand(m, generator=g, device=dev) < p).double() - 1.0
        return A, lab.cpu().numpy()
    # CSR: k uniformly drawn column indices per row, N(0,1) values, duplicates summed
    import scipy.sparse as sp

    k = cfg["nnz_per_row"]
    x_true = (torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.05)).double().cpu().numpy()
    g.manual_seed(1000 + rank)
    cols = torch.randint(0, n, (m, k), generator=g, device=dev, dtype=torch.int32)
    cols, _ = torch.sort(cols, dim=1)
    vals = torch.randn((m, k), generator=g, device=dev, dtype=torch.float32)
    noise = (0.1 * torch.randn(m, generator=g, device=dev)).double().cpu().numpy()
    ptr = np.arange(0, m * k + 1, k, dtype=np.int32)
    A = sp.csr_matrix((vals.cpu().numpy().ravel(), cols.cpu().numpy().ravel(), ptr), shape=(m, n))
    A.sum_duplicates()
    b = A @ x_true + noise
    return A, b


def functions(cfg, G, b, n):
    if cfg["kind"] == "dense_logistic":
        return G.logistic_functions(b, cfg["lambd"], n)
    return G.lasso_functions(b, cfg["lambd"], n)


def cpu_baseline(cfg, A_host, f, g, budget_s, engine, full_run=False):
    """Times the reference CPU path on this box's host cores, on the SAME (A, f, g).

    Dense: the compiled reference (oracle/_ref, `kind` "reference"; clean subprocess, it must
    not share a process with torch) -- first on a row sample that predicts the cost; if the
    full workload fits the budget it is run and reported together with `parity` (reference
    vs engine solution), otherwise the sample's it/s is scaled by the per-iteration byte
    ratio and the line says so.  Sparse (c4): the OpenMP oracle port (`kind` "port"; the
    reference's sparse path is single-threaded as built) on the leading rows (1/10 of them).
    Returns (cpu_baseline dict, parity dict or None)."""
    import numpy as np

    import oracle_binding as ob

    t_start = time.time()
    sparse = hasattr(A_host, "indptr")
    # threads actually used: the reference's BLAS threads (its best setting on this box, see
    # oracle_binding.REF_THREADS); the OpenMP oracle port uses every hardware thread
    cores = ob.cpu_quota() if sparse else ob.ref_threads()
    m, n = A_host.shape
    dt = np.float32
    soa = lambda fv, lo, hi: {k: getattr(fv, k)[lo:hi] for k in "habcde"}  # noqa: E731
    gs = soa(g, 0, n)

    def run(rows, use_ref, timeout):
        A = A_host[:rows]
        fs = soa(f, 0, rows)
        if use_ref:
            r = ob.ref_solve(A, fs, gs, dtype=dt, verbose=1, timeout=timeout)
            t_total, t_init = r.get("t_total", r["wall_s"]), r.get("t_init", 0.0)
        else:
            r = ob.oracle_solve(A, fs, gs, dtype=dt)
            t_init, t_total = r["info"]["t_init"], r["info"]["t_init"] + r["info"]["t_loop"]
        iters = r["iterations"] + 1
        ok = r["status"] == 0 and np.isfinite(r["optval"])
        return {"rows": rows, "iters": iters, "t_total": t_total, "t_init": t_init, "ok": ok,
                "its": iters / max(t_total - t_init, 1e-9), "res": r}

    out = {"unit": "it/s", "cores": cores, "host_threads_visible": os.cpu_count() or 1, "cpu_quota": ob.cpu_quota()}
    if sparse:
        rows = max(1, m // 10)
        s = run(rows, False, None)
        nnz_frac = A_host[:rows].nnz / max(A_host.nnz, 1)
        out.update(kind="port", value=s["its"] * nnz_frac,
                   sample="OpenMP oracle on the first %d rows (%d x %d, nnz %d): %d iterations, total %.1f s, init "
                          "%.1f s = %.2f it/s, scaled by the non-zero ratio %.3f to the full matrix"
                          % (rows, rows, n, A_host[:rows].nnz, s["iters"], s["t_total"], s["t_init"], s["its"], nnz_frac))
        return out, None
    kind = "reference" if ob.ref_available() else "port"
    # bounded sample: the leading 30 % of the rows (~20-30 s at C2 / C3 with 16 threads); the whole
    # workload only on request (--cpu-full)
    s_rows = min(m, max(2000, int(0.3 * m)))
    sample = None
    try:
        sample = run(s_rows, kind == "reference", budget_s)
        if not sample["ok"]:
            sample = None
    except Exception:
        sample = None
    if sample is None and kind == "reference":
        kind = "port"
        sample = run(s_rows, False, None)
    out["kind"] = kind
    remaining = budget_s - (time.time() - t_start)
    full = None
    if full_run and m > s_rows:
        try:
            full = run(m, kind == "reference", max(remaining, 3600.0))
            if not full["ok"]:
                full = None
        except Exception:
            full = None
    parity = None
    if full is not None:
        out.update(value=full["its"], time_to_converge_s=full["t_total"],
                   sample="full workload %dx%d fp32: %d iterations, total %.1f s, init %.1f s"
                          % (m, n, full["iters"], full["t_total"], full["t_init"]))
        r = full["res"]
        xr, xe = r["x"].astype(np.float64), engine["x"].astype(np.float64)
        parity = {"against": "compiled reference (oracle/_ref/libpogs_cpu.so), same A, b, lambda, default tolerances"
                  if kind == "reference" else "oracle port",
                  "rel_x": float(np.linalg.norm(xe - xr) / max(np.linalg.norm(xr), 1e-300)),
                  "rel_optval": abs(engine["optval"] - r["optval"]) / max(abs(r["optval"]), 1e-300),
                  "iterations_reference": full["iters"], "iterations_engine": engine["iterations"] + 1,
                  "tolerance": 1e-4}
    else:
        bytes_iter = lambda rows: 4.0 * (2.0 * rows * n + n * n)  # noqa: E731
        scale = bytes_iter(s_rows) / bytes_iter(m)
        out.update(value=sample["its"] * scale,
                   sample="first %d rows of the same A (%d iterations, total %.1f s, init %.1f s; %.2f it/s), "
                          "scaled by the per-iteration byte ratio %.3f to the %dx%d workload; the whole workload is "
                          "run with --cpu-full%s"
                          % (s_rows, sample["iters"], sample["t_total"], sample["t_init"], sample["its"], scale, m, n,
                             " (tests/golden/c2_reference.npz holds the reference's full-size C2 runs in the build container, "
                             "8 cores: fp32 154 iterations in 143 s = 2.6 it/s in the loop; fp64 106 iterations in 497 s)"
                             if cfg.get("cfg_index") == 1 else ""))
    return out, parity


def pmc_traffic(name, kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of
    this command (profiles/pmc_traffic_<config>.json: separate FETCH_SIZE / WRITE_SIZE passes,
    gfx950 half-count correction applied; scripts/pmc_summary.py).  (None, None) if absent."""
    rel = os.path.join("profiles", "pmc_traffic_%s.json" % name)
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, None
    try:
        d = json.load(open(path))
        sel = [e for k, e in d.items() if kernel_substr in k]
        cnt = sum(e["launches"] for e in sel)
        if not cnt:
            return None, None
        return sum(e["hbm_bytes_per_launch_corrected"] * e["launches"] for e in sel) / cnt, rel
    except Exception:
        return None, None


def main():
    args = parse()
    maybe_spawn(args)
    import oracle_binding as ob

    # host threads (OpenMP of the oracle port, torch's CPU ops): what the container may really use
    os.environ.setdefault("OMP_NUM_THREADS", str(ob.cpu_quota()))
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d"
              % (args.gpus, world, world), file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import pogs_amd
    from pogs_amd import graph as G

    cfg = CONFIGS[args.config]
    m = args.m or cfg["m"]
    n = args.n or cfg["n"]
    sparse = cfg["kind"] == "csr_lasso"

    dist_arg, dist = None, None
    force_dist = os.environ.get("POGS_AMD_FORCE_DIST", "0") == "1"  # exercise the RCCL path with 1 rank
    if world > 1 or force_dist:
        import torch.distributed as dist

        if force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl")
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid = torch.tensor(list(pogs_amd.dist_unique_id()), dtype=torch.uint8, device=dev)
        dist.broadcast(uid, 0)
        dist_arg = (rank, world, m * world, bytes(uid.cpu().tolist()))

    def barrier():
        torch.cuda.synchronize()
        if dist_arg is not None:
            dist.barrier()
        torch.cuda.synchronize()

    A, b = make_problem(cfg, m, n, rank, dev)
    torch.cuda.synchronize()
    t0 = time.time()
    if sparse:
        solver = pogs_amd.Solver(A, dtype=np.float32, device=local, profile=True, dist=dist_arg)
    else:
        from pogs_amd import _lib as L

        solver = pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True, device=local,
                                 profile=True, dist=dist_arg,
                                 projector=L.PROJ_CGLS if args.projector == "cgls" else L.PROJ_DEFAULT)
    init_s = time.time() - t0
    f, g = functions(cfg, G, b, n)

    # one complete cold solve: wall-clock-to-converge and the iteration count
    t0 = time.time()
    res = solver.solve(f, g)
    solve_s = time.time() - t0
    st_solve = solver.stats()

    solver.begin_run(f, g)
    solver.iterate(args.warmup)
    # exactly K steps between barriers; several windows when K steps are too short to time well
    per_step_guess = max(st_solve["t_loop_s"] / max(st_solve["iterations"], 1), 1e-6)
    windows = 1 if args.steps * per_step_guess >= 0.1 else min(25, max(3, int(0.25 / (args.steps * per_step_guess)) | 1))
    times = []
    solver.reset_stats()
    for _ in range(windows):
        barrier()
        t0 = time.time()
        solver.iterate(args.steps)
        barrier()
        elapsed = time.time() - t0
        if dist_arg is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        times.append(elapsed)
    elapsed = statistics.median(times)
    st = solver.stats()

    out_line = None
    if rank == 0:
        its = world * args.steps / elapsed
        launches = max(st["stream_launches"], 1)
        avg_ms = st["stream_ms"] / launches
        bytes_per_launch = st["stream_bytes"] / launches  # algorithmic, see DESIGN.md section 6
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        steps_total = args.steps * windows
        if sparse:
            nnz = A.nnz
            kernel = "spmv_sell_kernel (every SpMV of the loop: A p, A^T r, A x)"
            kernel_key = "spmv"
            spmv_per_iter = st["matvecs"] / max(steps_total, 1)
            iter_bytes = bytes_per_launch * spmv_per_iter
            iteration = {"bytes_model": "SpMVs per iteration x algorithmic bytes per SpMV (nnz (s + 4) + 4 (rows + 1) "
                                        "+ s (rows + cols), averaged over A and A^T)",
                         "spmv_per_iteration": spmv_per_iter, "cg_per_iteration": st["cg_iters"] / max(steps_total, 1),
                         "bytes": iter_bytes, "frac": iter_bytes * args.steps / elapsed / 1e9 / HBM_PEAK_GBS}
            workload = ("solve_lasso sparse CSR fp32 A=%dx%d nnz=%d per GPU, lambda=%g, default tolerances "
                        "(BASELINE.json configs[%d])" % (m, n, nnz, cfg["lambd"], cfg["cfg_index"]))
            projector = "CGLS (LDS-gather SpMV)"
        else:
            kernel = "stream_rows2_kernel<FusedIterOp> (the one pass over A per iteration)"
            kernel_key = "stream_rows2_kernel<float"
            one_pass = 4.0 * (m * n + 0.5 * n * n)  # A once + the lower triangle of W = L^-1
            two_pass = 4.0 * (2.0 * m * n + n * n)  # the reference algorithm
            iteration = {"bytes_model": "one-pass engine: A once + the lower triangle of W per iteration",
                         "bytes": one_pass, "frac": one_pass * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                         "passes_over_A_per_iteration": st["matvecs"] / max(steps_total, 1),
                         "vs_reference_algorithm_bytes": {
                             "bytes": two_pass, "note": "the reference reads A twice per iteration (+2 on exact-residual "
                                                        "iterations); this ratio is a speed-up over that byte model, NOT a "
                                                        "roofline fraction",
                             "ratio_to_hbm_peak": two_pass * args.steps / elapsed / 1e9 / HBM_PEAK_GBS}}
            name = "solve_lasso" if cfg["kind"] == "dense_lasso" else "solve_logistic"
            workload = ("%s dense fp32 A=%dx%d per GPU, lambda=%g, default tolerances (BASELINE.json configs[%d]%s)"
                        % (name, m, n, cfg["lambd"], cfg["cfg_index"],
                           "" if world == 1 else "; row-sharded %dx%d" % (m * world, n)))
            projector = "direct (MFMA Gram + Cholesky)"
            if args.projector == "cgls":
                passes = st["matvecs"] / max(steps_total, 1)
                kernel = "stream_rows_kernel (every pass over A of the loop: A p, A^T r, A x of CGLS and the residual passes)"
                kernel_key = "stream_rows_kernel<float"
                iteration = {"bytes_model": "passes over A per iteration x 4 m n bytes (matrix-free CGLS projector)",
                             "passes_over_A_per_iteration": passes, "cg_per_iteration": st["cg_iters"] / max(steps_total, 1),
                             "bytes": passes * 4.0 * m * n, "frac": passes * 4.0 * m * n * args.steps / elapsed / 1e9 / HBM_PEAK_GBS}
                projector = "CGLS on the dense matrix (matrix-free)"
                workload += " [--projector cgls]"
        traffic, traffic_src = (pmc_traffic(args.config, kernel_key) if (m, n) == (cfg["m"], cfg["n"]) and args.projector == "default"
                                else (None, None))
        line = {
            "metric": "admm_iterations_per_sec_%s_fp32 (per-GPU shard, summed over GPUs)" % cfg["kind"],
            "value": its, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "windows": windows, "window_s": times,
            "config": {"workload": workload, "name": args.config, "rows_per_gpu": m, "cols": n, "projector": projector,
                       "parallelism": "row-shard x%d" % world, "rccl_nranks": world if dist_arg is not None else 0},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("static: %s (rocprofv3 --pmc passes of this command, committed; not "
                                            "measured in this run)" % traffic_src) if traffic_src else None,
                         "kernel": kernel, "bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": avg_ms, "launches": st["stream_launches"], "iteration": iteration},
            "time_to_converge_s": init_s + solve_s, "init_s": init_s, "loop_s": st_solve["t_loop_s"],
            "solve_iterations": res["iterations"] + 1, "solve_status": res["status"],
            "exact_residual_iters": st_solve["exact_iters"],
            "setup_ms": {k: st_solve[k] for k in ("equil_ms", "normest_ms", "gram_ms", "chol_ms", "trtri_ms")},
        }
        if not sparse:
            line["gram_tflops"] = st_solve["gram_flops"] / max(st_solve["gram_ms"], 1e-9) / 1e9
        if world == 1 and not args.no_cpu_baseline:
            try:
                A_host = A if sparse else A.cpu().numpy()
                line["cpu_baseline"], parity = cpu_baseline(cfg, A_host, f, g, args.cpu_budget_s, res, args.cpu_full)
                if parity is not None:
                    line["parity_vs_reference"] = parity
            except Exception as e:  # the baseline must never take the bench line down
                line["cpu_baseline"] = {"value": None, "unit": "it/s", "cores": os.cpu_count(), "kind": "none",
                                        "sample": "failed: %r" % (e,)}
        out_line = json.dumps(line)
    solver.close()
    # The JSON line must be the LAST line of the job's stdout.  C libraries print through stdio
    # (RCCL's version banner: on a pipe it sits in the buffer until exit), so every rank empties
    # those buffers now, the ranks meet, and only then does rank 0 print.
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if dist_arg is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out_line is not None:
        print(out_line, flush=True)


if __name__ == "__main__":
    main()
