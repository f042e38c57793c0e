"""Benchmark of the hot path: ADMM iterations/s on the dense fp32 Lasso of BASELINE.json.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config C2 of SURVEY.md): solve_lasso, dense fp32, A = 100000 x 10000 per
GPU (synthetic N(0,1), x_true 10% dense, b = A x_true + 0.1 N(0,1), lambda = 0.1,
default tolerances).  With N GPUs the matrix is row-sharded, each rank holding
its own 100000 rows ("weak" scaling: N = 8 is config C5, 800000 x 10000); the
only per-iteration exchange is the RCCL all-reduce of the n-vector A_k^T y_k and
a few scalars.

A step is ONE ADMM iteration of a real default-tolerance solve (prox, gap and
tolerance sums, over-relaxation, projection = 2 passes over A + 2 triangular
products, residual bookkeeping, dual update, adaptive rho, and the exact-residual
pass whenever the reference would evaluate it); when a solve converges the next
step starts the next solve from the cold start, so K steps are K genuine
iterations.  The one-time setup (equilibration, norm estimate, MFMA Gram +
Cholesky) is outside the timed region and reported as init_s; a complete cold
solve is reported as time_to_converge_s.

value = N * K / T: iterations of one 100000 x 10000 shard per second, summed over
ranks (at N = 1 exactly the ADMM it/s of C2).  T is the max over ranks of the
time of exactly K steps between barrier + synchronize on both sides.

Inputs are resident in HBM when the timed region starts (A is generated on the
device).  The JSON line carries `roofline` (dominant kernel = the row-streaming
pass over A, timed with HIP events on the solver's stream over the timed region)
and, at N = 1, `cpu_baseline` (the compiled reference if oracle/_ref is loadable,
else the oracle restatement, on the host cores of this box).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

M_PER_GPU = 100000
N_COLS = 10000
LAMBDA = 0.1
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--m", type=int, default=M_PER_GPU, help="rows per GPU (default: the C2 shape)")
    ap.add_argument("--n", type=int, default=N_COLS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=90.0)
    return ap.parse_args()


def make_problem(m, n, rank, dev):
    """C2 generator on the device: per-rank rows, shared x_true."""
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    x_true = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.1)
    g.manual_seed(1000 + rank)
    A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
    b = A @ x_true + 0.1 * torch.randn(m, generator=g, device=dev)
    return A, b.double().cpu().numpy()


def cpu_baseline(A_host, b, n, budget_s):
    """Times the reference CPU path (or the oracle port) on this box's host cores.

    The compiled reference (oracle/_ref, `kind` "reference") runs in a clean
    subprocess (it must not share a process with torch, see oracle_binding.ref_solve);
    if it is unavailable or fails, the oracle restatement (`kind` "port") is timed.
    First a bounded sample (the first 20000 rows of the same A); if that predicts the
    full workload fits the remaining budget the full solve is run and reported,
    otherwise the sample's it/s is scaled by the per-iteration byte ratio."""
    import oracle_binding as ob
    from pogs_amd import graph as G

    cores = os.cpu_count() or 1
    t_start = time.time()

    def run(rows, use_ref, timeout):
        A = A_host[:rows]
        f, g = G.lasso_functions(b[:rows], LAMBDA, n)
        fs = {k: getattr(f, k) for k in "habcde"}
        gs = {k: getattr(g, k) for k in "habcde"}
        if use_ref:
            r = ob.ref_solve(A, fs, gs, dtype=np.float32, verbose=1, timeout=timeout)
            t_total, t_init = r.get("t_total", r["wall_s"]), r.get("t_init", 0.0)
        else:
            r = ob.oracle_solve(A, fs, gs, dtype=np.float32)
            t_init, t_total = r["info"]["t_init"], r["info"]["t_init"] + r["info"]["t_loop"]
        iters = r["iterations"] + 1
        ok = r["status"] == 0 and np.isfinite(r["optval"])
        return {"rows": rows, "iters": iters, "t_total": t_total, "t_init": t_init, "ok": ok,
                "its": iters / max(t_total - t_init, 1e-9)}

    m = A_host.shape[0]
    s_rows = min(m, 20000)
    kind, sample = "reference", None
    if ob.ref_available():
        try:
            sample = run(s_rows, True, budget_s)
            if not sample["ok"]:
                sample = None
        except Exception:
            sample = None
    if sample is None:
        kind = "port"
        sample = run(s_rows, False, None)
    bytes_iter = lambda rows: 4.0 * (2.0 * rows * n + n * n)  # noqa: E731
    out = {"unit": "it/s", "cores": cores, "kind": kind}
    predicted_full = sample["t_total"] * (m / s_rows) * 1.2
    remaining = budget_s - (time.time() - t_start)
    full = None
    if m > s_rows and predicted_full < remaining:
        try:
            full = run(m, kind == "reference", remaining)
            if not full["ok"]:
                full = None
        except Exception:
            full = None
    if full is not None:
        out.update(value=full["its"], time_to_converge_s=full["t_total"],
                   sample="full workload %dx%d fp32: %d iterations, total %.1f s, init %.1f s"
                          % (m, n, full["iters"], full["t_total"], full["t_init"]))
    else:
        scale = bytes_iter(s_rows) / bytes_iter(m)
        out.update(value=sample["its"] * scale,
                   sample="first %d rows of the same A (%d iterations, total %.1f s, init %.1f s; %.2f it/s), "
                          "scaled by the per-iteration byte ratio %.3f to the %dx%d workload"
                          % (s_rows, sample["iters"], sample["t_total"], sample["t_init"], sample["its"], scale,
                             m, n))
    return out


def pmc_traffic(m, n):
    """HBM bytes per launch of the A-streaming kernels from the committed rocprofv3 --pmc summary
    (profiles/pmc_traffic_c2.json: separate FETCH_SIZE / WRITE_SIZE passes of this same command,
    gfx950 half-count correction applied; scripts/pmc_summary.py).  None if it does not apply."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_c2.json")
    if (m, n) != (M_PER_GPU, N_COLS) or not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
        # launch-weighted mean over the kernels that stream A inside the ADMM loop
        sel = [e for k, e in d.items() if "stream_rows2_kernel<float" in k]
        n = sum(e["launches"] for e in sel)
        return sum(e["hbm_bytes_per_launch_corrected"] * e["launches"] for e in sel) / n if n else None
    except Exception:
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import pogs_amd
    from pogs_amd import graph as G

    dist_arg = None
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
        dist_arg = (rank, world, args.m * world, bytes(uid.cpu().tolist()))

    def barrier():
        torch.cuda.synchronize()
        if dist_arg is not None:
            dist.barrier()
        torch.cuda.synchronize()

    m, n = args.m, args.n
    A, b = make_problem(m, n, rank, dev)
    torch.cuda.synchronize()
    t0 = time.time()
    solver = pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True, device=local,
                             profile=True, dist=dist_arg)
    init_s = time.time() - t0
    f, g = G.lasso_functions(b, LAMBDA, n)

    # one complete cold solve: wall-clock-to-converge and the iteration count
    t0 = time.time()
    res = solver.solve(f, g)
    solve_s = time.time() - t0
    st_solve = solver.stats()

    solver.begin_run(f, g)
    solver.iterate(args.warmup)
    solver.reset_stats()
    barrier()
    t0 = time.time()
    solver.iterate(args.steps)
    barrier()
    elapsed = time.time() - t0
    if dist_arg is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = solver.stats()

    if rank == 0:
        its = world * args.steps / elapsed
        bytes_per_launch = 4.0 * m * n
        avg_ms = st["stream_ms"] / max(st["stream_launches"], 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        iter_bytes = 4.0 * (2.0 * m * n + n * n)  # algorithmic bytes per iteration
        line = {
            "metric": "admm_iterations_per_sec_dense_lasso_fp32 (per 100000x10000 shard, summed over GPUs)",
            "value": its, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "solve_lasso dense fp32 A=%dx%d per GPU, lambda=0.1, default tolerances "
                                   "(BASELINE.json configs[1]%s)" % (m, n, "" if world == 1 else
                                                                     "; row-sharded %dx%d" % (m * world, n)),
                       "rows_per_gpu": m, "cols": n, "projector": "direct (MFMA Gram + Cholesky)",
                       "parallelism": "row-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(m, n),
                         "kernel": "stream_rows2_kernel<FusedIterOp> (the one pass over A per iteration)", "bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": avg_ms, "launches": st["stream_launches"],
                         "iteration_frac": iter_bytes * args.steps / elapsed / 1e9 / HBM_PEAK_GBS},
            "time_to_converge_s": init_s + solve_s, "init_s": init_s, "loop_s": st_solve["t_loop_s"],
            "solve_iterations": res["iterations"] + 1, "solve_status": res["status"],
            "exact_residual_iters": st_solve["exact_iters"],
            "setup_ms": {k: st_solve[k] for k in ("equil_ms", "normest_ms", "gram_ms", "chol_ms", "trtri_ms")},
            "gram_tflops": st_solve["gram_flops"] / max(st_solve["gram_ms"], 1e-9) / 1e9,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(A.cpu().numpy(), b, n, args.cpu_budget_s)
            except Exception as e:  # the baseline must never take the bench line down
                line["cpu_baseline"] = {"value": None, "unit": "it/s", "cores": os.cpu_count(), "kind": "none",
                                        "sample": "failed: %r" % (e,)}
        print(json.dumps(line))
    solver.close()
    if dist_arg is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
