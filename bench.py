"""Benchmark of the hot path: ADMM iterations/s of the graph-form solve on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4]

With N > 1 and no launcher in the environment the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`
(one rank per GPU, RCCL over xGMI); under a launcher (RANK / WORLD_SIZE set) it runs as
that rank.

Workloads (BASELINE.json `configs`); rows are PER GPU ("weak" scaling):
  c2 (default, the metric's configuration): solve_lasso, dense fp32, A = 100000 x 10000 per
      GPU, synthetic N(0,1), x_true 10% dense, b = A x_true + 0.1 N(0,1), lambda = 0.1,
      default tolerances, direct projector.  N = 8 is config C5 (800000 x 10000).  At N = 1 the
      matrix is numpy's (pogs_amd.synth.dense_lasso_rows(seed=2024), the committed fixture's
      problem), at N > 1 every rank draws its rows on its device (torch, seed 1000 + rank).
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
windows are timed back to back and their mean is reported (`windows`, `window_s`).

Inputs are resident in HBM when the timed region starts.  The JSON line carries `roofline`
(the dominant kernel, timed with HIP events on the solver's stream over the timed region)
and, at N = 1, `cpu_baseline`: the compiled reference (oracle/_ref, clean subprocess) on the
same (A, b, lambda) on this box's host cores, on the WHOLE matrix (no row sample, nothing
scaled), capped at --cpu-iters ADMM iterations -- on the GPU box's host its fp32 build does not
reach the default tolerances at c2 and would run twenty minutes into max_iter
(profiles/r03_ref_cpu_diagnosis.md); --cpu-full runs it to its end.  `parity_vs_reference`
for c2 is taken against the committed reference solutions of exactly the benchmarked problem
(tests/golden/c2_reference.npz: at one GPU c2's matrix is that fixture's, regenerated from its
seed).  c4: the OpenMP oracle port on the whole workload to convergence.
Without --config (the driver's invocation) and at N = 1 the c3 and c4 workloads are run after c2
(GPU legs only) and attached as `secondary`: {c3: {...}, c4: {...}} with their own value /
ms_per_step / roofline.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PROFILE_EVERY = 4   # HIP-event brackets on every 4th launch of the dominant kernel (a bracket costs the stream ~6 us)
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
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="default: c2 as the headline line, and at --gpus 1 also c3 and c4 as `secondary`")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--m", type=int, default=0, help="rows per GPU (default: the configuration's)")
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--projector", choices=["default", "cgls"], default="default",
                    help="dense configurations: 'cgls' selects the matrix-free CGLS projector (the reference's "
                         "ProjectorCgls on a dense matrix, src/cpu/projector/projector_cgls.cpp) instead of the direct one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=600.0,
                    help="time-out of the reference's run of the whole workload (the CPU baseline)")
    ap.add_argument("--cpu-iters", type=int, default=60, help="ADMM iterations of the reference's whole-workload run")
    ap.add_argument("--cpu-full", action="store_true", help="run the reference to convergence instead (c2 on the GPU "
                                                            "box's host: max_iter, about twenty minutes)")
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


def fixture_c2():
    path = os.path.join(ROOT, "tests", "golden", "c2_reference.npz")
    if not os.path.exists(path):
        return None
    import numpy as np

    return np.load(path)


def make_problem(cfg, m, n, rank, dev, world=1):
    """Per-rank rows with a shared x_true / w_true (generated on the device).  Returns
    (matrix: device tensor or scipy CSR, b or labels, host copy of a dense matrix or None).

    c2 on ONE GPU at its own size is the problem of the committed fixture
    tests/golden/c2_reference.npz (pogs_amd.synth.dense_lasso_rows(seed=2024), regenerated bit for
    bit on the host and checked against the fixture's checksums): the compiled reference's fp32
    and fp64 solutions of exactly this (A, b, lambda) are in the fixture, so the line carries
    `parity_vs_reference` without a 20-minute reference run on this box."""
    import numpy as np
    import torch

    fx = fixture_c2() if (cfg["kind"] == "dense_lasso" and world == 1) else None
    if fx is not None and (m, n) == tuple(int(v) for v in fx["shape"]):
        from pogs_amd import synth

        A_host, b, _ = synth.dense_lasso_rows(m, n, seed=int(fx["seed"]))
        chk = np.array([float(A_host[::997].astype(np.float64).sum()), float(np.abs(A_host[:, ::113]).astype(np.float64).sum()),
                        float(np.linalg.norm(b)), float(b[::101].sum())])
        if not np.allclose(chk, fx["checksums"], rtol=1e-12, atol=0):
            raise RuntimeError("the generator no longer reproduces the fixture's inputs")
        return torch.from_numpy(A_host).to(dev), b, A_host
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    kind = cfg["kind"]
    if kind == "dense_lasso":
        x_true = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.1)
        g.manual_seed(1000 + rank)
        A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
        b = A @ x_true + 0.1 * torch.randn(m, generator=g, device=dev)
        return A, b.double().cpu().numpy(), None
    if kind == "dense_logistic":
        w = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.3)
        w = w * (2.0 / torch.sqrt((w * w).sum()))
        g.manual_seed(1000 + rank)
        A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
        p = torch.sigmoid(A @ w)
        lab = 2.0 * (torch.rand(m, generator=g, device=dev) < p).double() - 1.0
        return A, lab.cpu().numpy(), None
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
    return A, b, None


def functions(cfg, G, b, n):
    if cfg["kind"] == "dense_logistic":
        return G.logistic_functions(b, cfg["lambd"], n)
    return G.lasso_functions(b, cfg["lambd"], n)


def cpu_baseline(cfg, A_host, f, g, args, engine, fixture=None):
    """Times the reference CPU path on this box's host cores, on the SAME (A, f, g), WHOLE workload:
    no row sample, nothing scaled.

    Dense (c2, c3): the compiled reference (oracle/_ref, `kind` "reference"; a clean subprocess, it
    must not share a process with torch) on all rows; `value` = iterations / (Total - Init) from its
    own summary line (src/cpu/pogs.cpp:485-490).  The run is capped at --cpu-iters ADMM iterations
    (default 60): on the GPU box's host the reference's fp32 build does not reach the default
    tolerances at c2 -- its primal residual stalls 0.2 % above the bound from iteration ~200 on and
    it runs into max_iter = 2500, twenty minutes (profiles/r03_ref_cpu_diagnosis.md) -- so a run to
    convergence is only made on request (--cpu-full).  The capped run is the reference's own loop on
    the whole matrix; its first iterations are its cheapest (no exact-residual passes yet), so the
    cap flatters the CPU side, not the GPU side.
    `parity`: against the committed reference solutions of exactly this problem when the workload is
    the fixture's (c2, one GPU); against the live run when that ran to convergence.
    Sparse (c4): the OpenMP oracle port on the whole workload to convergence (`kind` "port": the
    reference's sparse path is single-threaded as built and needs 21 minutes,
    tests/golden/make_c4_reference.py).  Returns (cpu_baseline dict, parity dict or None)."""
    import numpy as np

    import oracle_binding as ob

    t_start = time.time()
    sparse = hasattr(A_host, "indptr")
    m, n = A_host.shape
    dt = np.float32
    soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
    fs, gs = soa(f), soa(g)

    def parity_of(x_ref, optval_ref, it_ref, against):
        xr, xe = np.asarray(x_ref, np.float64), engine["x"].astype(np.float64)
        return {"against": against, "rel_x": float(np.linalg.norm(xe - xr) / max(np.linalg.norm(xr), 1e-300)),
                "rel_optval": abs(engine["optval"] - optval_ref) / max(abs(optval_ref), 1e-300),
                "iterations_reference": int(it_ref), "iterations_engine": engine["iterations"] + 1, "tolerance": 1e-4}

    out = {"unit": "it/s", "host_threads_visible": os.cpu_count() or 1, "cpu_quota": ob.cpu_quota()}
    parity = None
    if fixture is not None:
        parity = parity_of(fixture["x_fp64"], float(fixture["optval_fp64"]), int(fixture["iterations_fp64"]) + 1,
                           "tests/golden/c2_reference.npz: the compiled reference's fp64 build (PogsD) on exactly this (A, b, "
                           "lambda), run to convergence in the build container")
        p32 = parity_of(fixture["x"], float(fixture["optval"]), int(fixture["iterations"]) + 1, "")
        parity["rel_x_vs_reference_fp32_build"] = p32["rel_x"]
        parity["iterations_reference_fp32_build"] = p32["iterations_reference"]
    if sparse:
        ob.oracle_set_threads()
        r = ob.oracle_solve(A_host, fs, gs, dtype=dt)
        t_init, t_loop = r["info"]["t_init"], r["info"]["t_loop"]
        iters = r["iterations"] + 1
        out.update(kind="port", cores=ob.cpu_quota(), value=iters / max(t_loop, 1e-9), time_to_converge_s=t_init + t_loop,
                   sample="whole workload %dx%d nnz %d fp32 to convergence, OpenMP oracle port (oracle/pogs_oracle.cpp, pinned to "
                          "the reference in tests/), %d threads: %d iterations, total %.1f s, init %.1f s"
                          % (m, n, A_host.nnz, ob.cpu_quota(), iters, t_init + t_loop, t_init))
        return out, parity_of(r["x"], r["optval"], iters, "oracle port, same A, b, lambda, default tolerances, whole workload")
    if not ob.ref_available():
        raise RuntimeError("oracle/_ref/libpogs_cpu.so is missing (built by __graft_entry__.build() where /root/reference exists)")
    cap = None if args.cpu_full else args.cpu_iters
    r = ob.ref_solve(A_host, fs, gs, dtype=dt, verbose=1, timeout=args.cpu_budget_s, **({"max_iter": cap} if cap else {}))
    t_total, t_init = r.get("t_total", r["wall_s"]), r.get("t_init", 0.0)
    iters = r["iterations"] + 1
    converged = r["status"] == 0
    out.update(kind="reference", cores=ob.ref_threads(), value=iters / max(t_total - t_init, 1e-9), init_s=t_init,
               loop_s=t_total - t_init, iterations=iters, converged=converged, threads_env=ob.ref_env_note(),
               sample="whole workload %dx%d fp32, compiled reference (oracle/_ref/libpogs_cpu.so), %s: %d iterations%s, "
                      "Total %.1f s, Init %.1f s (its own summary line); elapsed with process start and input hand-over %.1f s"
                      % (m, n, "to convergence" if converged else "max_iter = %d" % (cap or 2500), iters,
                         "" if converged else " (not converged: capped, see cpu_baseline() in bench.py)", t_total, t_init,
                         time.time() - t_start))
    if converged:
        out["time_to_converge_s"] = t_total
        if parity is None:
            parity = parity_of(r["x"], r["optval"], iters, "compiled reference (oracle/_ref/libpogs_cpu.so), same A, b, lambda, "
                                                           "default tolerances, whole workload, this run")
    return out, parity


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources, as scripts/pmc_summary.py records it."""
    import hashlib

    d = os.path.join(ROOT, "pogs_amd", "csrc")
    hh = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".h", ".hip")):
            hh.update(fn.encode())
            hh.update(open(os.path.join(d, fn), "rb").read())
    return hh.hexdigest()[:16]


def pmc_traffic(name, kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of
    this command (profiles/pmc_traffic_<config>.json: separate FETCH_SIZE / WRITE_SIZE passes,
    gfx950 half-count correction applied; scripts/pmc_summary.py).  Returns (bytes, file, fresh):
    `fresh` says whether the counters were collected on the kernel sources this run uses (the file
    records their hash).  (None, None, None) if absent."""
    rel = os.path.join("profiles", "pmc_traffic_%s.json" % name)
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, None, None
    try:
        d = json.load(open(path))
        meta = d.pop("_meta", {})
        sel = [e for k, e in d.items() if kernel_substr in k]
        cnt = sum(e["launches"] for e in sel)
        if not cnt:
            return None, None, None
        fresh = meta.get("csrc_sha16") == csrc_sha16() if meta.get("csrc_sha16") else None
        return sum(e["hbm_bytes_per_launch_corrected"] * e["launches"] for e in sel) / cnt, rel, fresh
    except Exception:
        return None, None, None


class Env:
    """What every configuration of a run shares: ranks, device, the process group."""

    def __init__(self, args):
        import torch

        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus and self.rank == 0:
            print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d"
                  % (args.gpus, self.world, self.world), file=sys.stderr)
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        self.dist = None
        self.force_dist = os.environ.get("POGS_AMD_FORCE_DIST", "0") == "1"  # exercise the RCCL path with 1 rank
        if self.world > 1 or self.force_dist:
            import torch.distributed as dist

            if self.force_dist and "MASTER_ADDR" not in os.environ:
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
            dist.init_process_group("nccl")
            self.dist = dist

    def barrier(self):
        import torch

        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()


def run_config(env, name, with_cpu):
    """One configuration: build, one cold solve, the timed windows.  Returns the JSON dict on rank 0
    (None elsewhere)."""
    import numpy as np
    import torch

    import pogs_amd
    from pogs_amd import graph as G

    args, rank, world, local, dev, dist = env.args, env.rank, env.world, env.local, env.dev, env.dist
    cfg = CONFIGS[name]
    m = args.m or cfg["m"]
    n = args.n or cfg["n"]
    sparse = cfg["kind"] == "csr_lasso"

    def new_dist_arg():
        """(rank, world, global rows, a FRESH RCCL unique id from rank 0): one per solver handle."""
        if dist is None:
            return None
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid = torch.tensor(list(pogs_amd.dist_unique_id()), dtype=torch.uint8, device=dev)
        dist.broadcast(uid, 0)
        return (rank, world, m * world, bytes(uid.cpu().tolist()))

    A, b, A_host = make_problem(cfg, m, n, rank, dev, world)
    torch.cuda.synchronize()

    if sparse:
        # the CSR arrays resident in HBM, like the dense matrices: the timed setup starts from there
        csr_dev = (torch.from_numpy(np.ascontiguousarray(A.data, np.float32)).to(dev),
                   torch.from_numpy(np.ascontiguousarray(A.indptr, np.int32)).to(dev),
                   torch.from_numpy(np.ascontiguousarray(A.indices, np.int32)).to(dev))
        torch.cuda.synchronize()

    def create():
        dist_arg = new_dist_arg()
        if sparse:
            return pogs_amd.Solver((csr_dev[0].data_ptr(), csr_dev[1].data_ptr(), csr_dev[2].data_ptr(), A.nnz),
                                   dtype=np.float32, shape=(m, n), device_ptr=True, device=local, profile=PROFILE_EVERY,
                                   dist=dist_arg)
        from pogs_amd import _lib as L

        return pogs_amd.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True, device=local,
                               profile=PROFILE_EVERY, dist=dist_arg,
                               projector=L.PROJ_CGLS if args.projector == "cgls" else L.PROJ_DEFAULT)

    # The one-time setup is timed twice: the first handle of a process also pays for the HIP stream and the
    # host-mapped scalar page (6 ms) and for the runtime loading every code object on its first launch
    # (~3 ms per MB); the second one -- what a process that has solved anything before sees -- is init_s.
    t0 = time.time()
    solver = create()
    init_cold_s = time.time() - t0
    solver.close()
    env.barrier()
    t0 = time.time()
    solver = create()
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
        env.barrier()
        t0 = time.time()
        solver.iterate(args.steps)
        env.barrier()
        elapsed = time.time() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        times.append(elapsed)
    # the windows are consecutive stretches of the same real solves: their MEAN is the iteration rate
    # (the median would drop the windows that hold a solve's dearer iterations -- a missed rho
    # prediction's extra pass, the two-step CG projections late in a sparse solve)
    elapsed = sum(times) / len(times)
    st = solver.stats()
    nranks_comm = st.get("comm_nranks", 0)   # as ncclCommCount reports it (0: no communicator)

    line = None
    if rank == 0:
        its = world * args.steps / elapsed
        launches = max(st["stream_launches"], 1)
        avg_ms = st["stream_ms"] / launches
        bytes_per_launch = st["stream_bytes"] / launches  # algorithmic, see DESIGN.md section 6
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        steps_total = args.steps * windows
        if sparse:
            nnz = A.nnz
            kernel = "spmv_sell_fin_kernel / spmv_sell_kernel (every SpMV of the loop: A p, A^T r, A x)"
            kernel_key = "spmv"
            spmv_per_iter = st["matvecs"] / max(steps_total, 1)
            iter_bytes = bytes_per_launch * spmv_per_iter
            iteration = {"bytes_model": "SpMVs per iteration x algorithmic bytes per SpMV (nnz (s + 4) + 4 (rows + 1) "
                                        "+ s (rows + cols), averaged over A and A^T)",
                         "spmv_per_iteration": spmv_per_iter, "cg_per_iteration": st["cg_iters"] / max(steps_total, 1),
                         "bytes": iter_bytes, "frac": iter_bytes * args.steps / elapsed / 1e9 / HBM_PEAK_GBS}
            workload = ("solve_lasso sparse CSR fp32 A=%dx%d nnz=%d per GPU, lambda=%g, default tolerances "
                        "(BASELINE.json configs[%d])" % (m, n, nnz, cfg["lambd"], cfg["cfg_index"]))
            projector = "CGLS (LDS-gather SpMV, device-resident CG loop)"
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
            wname = "solve_lasso" if cfg["kind"] == "dense_lasso" else "solve_logistic"
            workload = ("%s dense fp32 A=%dx%d per GPU, lambda=%g, default tolerances (BASELINE.json configs[%d]%s)"
                        % (wname, m, n, cfg["lambd"], cfg["cfg_index"],
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
        traffic, traffic_src, traffic_fresh = (pmc_traffic(name, kernel_key)
                                               if (m, n) == (cfg["m"], cfg["n"]) and args.projector == "default"
                                               else (None, None, None))
        line = {
            "metric": "admm_iterations_per_sec_%s_fp32 (per-GPU shard, summed over GPUs)" % cfg["kind"],
            "value": its, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "windows": windows, "window_s": times,
            "config": {"workload": workload, "name": name, "rows_per_gpu": m, "cols": n, "projector": projector,
                       "parallelism": "row-shard x%d" % world, "rccl_nranks": nranks_comm},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("static: %s (rocprofv3 --pmc passes of this command, committed; not "
                                            "measured in this run; %s)"
                                            % (traffic_src, "collected on the kernel sources of this build" if traffic_fresh
                                               else "STALE: the kernel sources have changed since the counters were collected"
                                               if traffic_fresh is False else "no source hash in the file")) if traffic_src else None,
                         "kernel": kernel, "bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": avg_ms, "launches": st["stream_launches"],
                         "launch_sampling": "HIP events around every %d-th launch of the kernel in the timed region" % PROFILE_EVERY,
                         "iteration": iteration},
            "time_to_converge_s": init_s + solve_s, "init_s": init_s, "loop_s": st_solve["t_loop_s"],
            "first_handle_of_the_process": {"init_s": init_cold_s, "time_to_converge_s": init_cold_s + solve_s,
                                            "note": "the process's first solver handle also pays for HIP stream creation and "
                                                    "code-object loading; init_s / time_to_converge_s are a second handle's"},
            "solve_iterations": res["iterations"] + 1, "solve_status": res["status"],
            "exact_residual_iters": st_solve["exact_iters"],
            "setup_ms": {k: st_solve[k] for k in ("equil_ms", "normest_ms", "gram_ms", "chol_ms", "trtri_ms")},
        }
        if not sparse:
            line["gram_tflops"] = st_solve["gram_flops"] / max(st_solve["gram_ms"], 1e-9) / 1e9
    solver.close()
    if rank == 0 and with_cpu:
        try:
            if sparse:
                A_host = A
            elif A_host is None:
                A_host = A.cpu().numpy()
            del A   # the GPU copy is not needed any more; the reference gets the host copy
            torch.cuda.empty_cache()
            fx = fixture_c2() if (name == "c2" and (m, n) == (cfg["m"], cfg["n"])) else None
            line["cpu_baseline"], parity = cpu_baseline(cfg, A_host, f, g, args, res, fx)
            if parity is not None:
                line["parity_vs_reference"] = parity
        except Exception as e:  # the baseline must never take the bench line down
            line["cpu_baseline"] = {"value": None, "unit": "it/s", "cores": os.cpu_count(), "kind": "none",
                                    "sample": "failed: %r" % (e,)}
    return line


def main():
    args = parse()
    maybe_spawn(args)
    import oracle_binding as ob

    # host threads (OpenMP of the oracle port, torch's CPU ops): what the container may really use
    os.environ.setdefault("OMP_NUM_THREADS", str(ob.cpu_quota()))
    env = Env(args)
    head = args.config or "c2"
    line = run_config(env, head, with_cpu=env.world == 1 and not args.no_cpu_baseline)
    # the driver's invocation (no --config, one GPU): c3 and c4 under the same clock, GPU legs only
    if args.config is None and env.world == 1 and not args.no_secondary and not (args.m or args.n) \
            and args.projector == "default":
        keep = ("value", "unit", "ms_per_step", "steps", "windows", "window_s", "config", "roofline", "time_to_converge_s", "init_s",
                "solve_iterations", "solve_status", "setup_ms")
        sec = {}
        for name in ("c3", "c4"):
            try:
                d = run_config(env, name, with_cpu=False)
                sec[name] = {k: d[k] for k in keep if k in d}
            except Exception as e:
                sec[name] = {"value": None, "error": repr(e)[:300]}
        if line is not None:
            line["secondary"] = sec
    out_line = json.dumps(line) if line is not None else None
    # The JSON line must be the LAST line of the job's stdout.  C libraries print through stdio
    # (RCCL's version banner: on a pipe it sits in the buffer until exit), so every rank empties
    # those buffers now, the ranks meet, and only then does rank 0 print.
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if env.dist is not None:
        env.dist.barrier()
        env.dist.destroy_process_group()
    if out_line is not None:
        print(out_line, flush=True)


if __name__ == "__main__":
    main()
