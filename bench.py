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
  c2f64: c2's matrix widened to fp64 -- the arithmetic type of the reference's Python layer
      (python/pogs/graph.py:281-288), i.e. what a caller who keeps its calling convention runs.

At N = 1 every workload is the problem of a committed fixture of the COMPILED REFERENCE's own
solution (tests/golden/c2_reference.npz, c3_reference.npz, c4_reference.npz: numpy PCG64
generators of pogs_amd/synth.py, regenerated here bit for bit and checked against the fixture's
checksums), so each line carries `parity_vs_reference`.  At N > 1 every rank draws its rows on its
device (torch, seed 1000 + rank; x_true shared): the shards are row ranges of ONE problem, which
rank 0 regenerates whole and -- when it fits one GPU -- solves unsharded after the timed region;
`parity_vs_reference` then holds the sharded solution against that solve (the unsharded engine is
the one pinned to the reference at N = 1).

A step is ONE ADMM iteration of a real default-tolerance solve (prox, gap and tolerance
sums, over-relaxation, projection, residual bookkeeping, dual update, adaptive rho, exact
residuals whenever the reference would evaluate them); when a solve converges the next step
starts the next solve from the cold start, so K steps are K genuine iterations.  The one-time
setup (equilibration, norm estimate, Gram + Cholesky / blocked SpMV layout) is outside the
timed region (init_s); a complete cold solve is reported as time_to_converge_s, and
`handle_cycles` repeats create / solve / destroy five times (what a caller of the one-shot ABI
pays per call, src/interface_c/pogs_c.cpp:19-20) and reports every cycle and the slowest.

value = N * K / T: iterations of one per-GPU shard per second, summed over ranks (at N = 1 the
ADMM it/s of the configuration).  T is the max over ranks of the time of exactly K steps
between barrier + synchronize on both sides; as many such windows are timed back to back as make
the timed stretch a WHOLE NUMBER OF SOLVES (`windows`, `window_s`, `windows_cover`; pick_windows) and
their mean is reported: the iterations of a solve do not all cost the same (a sparse solve's CGLS
projections take 4 steps early, 1 in the middle, 2-3 at the end), and the metric is a solve's
iterations over its loop time.

Inputs are resident in HBM when the timed region starts.  The JSON line carries `roofline`
(the dominant kernel, timed with HIP events on the solver's stream over the timed region;
`roofline.traffic` of the headline workload is MEASURED BY THIS RUN at N = 1: after everything
timed is done the script runs itself twice more for a few steps under `rocprofv3 --pmc
FETCH_SIZE` / `--pmc WRITE_SIZE` -- about a minute, --no-live-traffic skips it -- and the
committed counter summaries profiles/pmc_traffic_<config>.json stay in the line as
`traffic_static`; the secondary workloads quote their committed summaries)
and `cpu_baseline`: the reference CPU path on the same (A, b, lambda) on this box's
host cores, WHOLE workload (no row sample, nothing scaled) -- at N > 1 rank 0 times it on its OWN
shard after the timed region (the other ranks wait on sockets) and the line gives the whole problem's
rate as `value / N`, marked `extrapolated` (every per-iteration term of the reference is linear in m; the
whole C5 problem is beyond its int-sized views), and `roofline.traffic` is measured on rank 0's GPU.
The line also carries both roofline peaks (`peak_datasheet`, `peak_measured`: the library's read probe),
`exact_setup` (the same create + solve with the setup's two shortcuts off), `one_shot_host_call` (the
reference's entry point with a HOST matrix, and the upload on its own).

OUTPUT: two lines on stdout.  First `BENCH_DETAIL {...}`: the long record, every workload's whole dictionary
(also written to gpurun_out/bench_detail.json).  Then, LAST, the contract line (summary_line): starts with
{"metric", at most 6000 characters, with `roofline`, `cpu_baseline`, the wall-clock and parity scalars of the
headline workload and a short block (value, roofline, parity, CPU figure) per secondary workload -- the driver
parses the last line out of an 8 KB tail of stdout.

CPU leg.  Dense: the compiled reference in
both BLAS builds -- oracle/_ref/libpogs_cpu_openblas.so (scipy's OpenBLAS, which threads its
gemv: the "best configuration") and oracle/_ref/libpogs_cpu.so (MKL, the build the oracle is
pinned to; on the GPU box's AMD host its sgemv runs on one thread,
profiles/r03_ref_cpu_diagnosis.md) -- c2 capped at --cpu-iters ADMM iterations (the fp32 build
does not reach the default tolerances there and would run twenty minutes into max_iter;
--cpu-full runs it to its end), c3 to convergence.  c4: the OpenMP oracle port to convergence
(the reference's sparse path is single-threaded: 21 minutes).
Without --config (the driver's invocation) and at N = 1 the c3, c4 and c2f64 workloads are run
after c2 and attached as `secondary`, each with its own value / roofline / parity (c3, c4 also
with their CPU leg).
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
    "c2": dict(m=100000, n=10000, kind="dense_lasso", lambd=0.1, cfg_index=1, dtype="f32", fixture="c2_reference.npz"),
    "c3": dict(m=200000, n=5000, kind="dense_logistic", lambd=0.01, cfg_index=2, dtype="f32", fixture="c3_reference.npz"),
    "c4": dict(m=2000000, n=500000, kind="csr_lasso", lambd=0.1, nnz_per_row=50, cfg_index=3, dtype="f32",
               fixture="c4_reference.npz"),
    "c2f64": dict(m=100000, n=10000, kind="dense_lasso", lambd=0.1, cfg_index=1, dtype="f64", fixture="c2_reference.npz"),
}
HANDLE_CYCLES = 5
UNSHARDED_CHECK_MAX_BYTES = 96e9   # N > 1: rank 0 solves the whole problem unsharded when the matrix is at most this


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="default: c2 as the headline line, and at --gpus 1 also c3 and c4 as `secondary`")
    ap.add_argument("--no-secondary", action="store_true")
    # (--rows-per-gpu / --cols: under `python -m torch.distributed.run ... bench.py ARGS` the launcher's own parser
    # reads a bare --m or --n as an ambiguous abbreviation of ITS options -- found by the two-process rehearsal)
    ap.add_argument("--m", "--rows-per-gpu", dest="m", type=int, default=0, help="rows per GPU (default: the configuration's)")
    ap.add_argument("--n", "--cols", dest="n", type=int, default=0)
    ap.add_argument("--projector", choices=["default", "cgls"], default="default",
                    help="dense configurations: 'cgls' selects the matrix-free CGLS projector (the reference's "
                         "ProjectorCgls on a dense matrix, src/cpu/projector/projector_cgls.cpp) instead of the direct one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two short rocprofv3 --pmc passes of the headline "
                         "workload, spawned after the timed part); the committed counter summary is quoted instead")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)   # the spawned pass itself
    ap.add_argument("--cpu-budget-s", type=float, default=600.0,
                    help="time-out of the reference's run of the whole workload (the CPU baseline)")
    ap.add_argument("--cpu-iters", type=int, default=60, help="ADMM iterations of the reference's whole-workload run")
    ap.add_argument("--cpu-best-only", action="store_true",
                    help="time only the best BLAS build of the reference (what an N > 1 run does while the other ranks wait)")
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


def load_fixture(cfg):
    path = os.path.join(ROOT, "tests", "golden", cfg["fixture"])
    if not os.path.exists(path):
        return None
    import numpy as np

    return np.load(path)


def torch_rows(cfg, m, n, rank, dev):
    """Rank `rank`'s m rows of the N > 1 problem (and of custom sizes): x_true / w_true from seed 1234
    (shared), rows and noise from seed 1000 + rank, all drawn on the device.  The same call on another
    device of the same kind gives the same bits, which is how rank 0 regenerates the whole problem."""
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
    return A, A @ x_true + noise


def make_problem(cfg, m, n, rank, dev, world=1):
    """Returns (matrix: device tensor or scipy CSR, b or labels, host copy of a dense matrix or None,
    fixture or None).

    ONE GPU at the configuration's own size: the problem of the committed fixture of the compiled
    reference's solution (tests/golden/<cfg["fixture"]>), regenerated bit for bit on the host with
    the numpy generators of pogs_amd/synth.py and checked against the fixture's checksums -- so the
    line carries `parity_vs_reference` without a reference run of minutes (c2, c3) to an hour (c4) on
    this box.  Otherwise: torch_rows."""
    import numpy as np
    import torch

    from pogs_amd import synth

    fx = load_fixture(cfg) if world == 1 else None
    if fx is not None and (m, n) == tuple(int(v) for v in fx["shape"][:2]):
        kind = cfg["kind"]
        if kind == "csr_lasso":
            A, b, _ = synth.csr_lasso(m, n, cfg["nnz_per_row"], seed=int(fx["seed"]), dtype=np.float32)
            chk = np.array([float(A.nnz), float(A.data[::1009].astype(np.float64).sum()),
                            float(A.indices[::1013].astype(np.float64).sum()), float(np.linalg.norm(b)), float(b[::101].sum())])
            A_host = None
        else:
            if kind == "dense_lasso":
                A_host, b, _ = synth.dense_lasso_rows(m, n, seed=int(fx["seed"]))
            else:
                A_host, b, _ = synth.dense_logistic_rows(m, n, seed=int(fx["seed"]), logit_std=float(fx["logit_std"]))
            chk = np.array([float(A_host[::997].astype(np.float64).sum()), float(np.abs(A_host[:, ::113]).astype(np.float64).sum()),
                            float(np.linalg.norm(b)) if kind == "dense_lasso" else float(b.sum()), float(b[::101].sum())])
            A = torch.from_numpy(A_host).to(dev)
            if cfg["dtype"] == "f64":
                A = A.double()      # widened on the device: the entries are the fp32 matrix's, exactly
                A_host = None       # (no CPU leg for this workload)
        if not np.allclose(chk, fx["checksums"], rtol=1e-12, atol=0):
            raise RuntimeError("the generator no longer reproduces the inputs of %s" % cfg["fixture"])
        return A, b, A_host, fx
    A, b = torch_rows(cfg, m, n, rank, dev)
    if cfg["dtype"] == "f64" and cfg["kind"] != "csr_lasso":
        A = A.double()
    return A, b, None, None


def functions(cfg, G, b, n):
    if cfg["kind"] == "dense_logistic":
        return G.logistic_functions(b, cfg["lambd"], n)
    return G.lasso_functions(b, cfg["lambd"], n)


def _parity(engine, x_ref, optval_ref, it_ref, against):
    import numpy as np

    xr, xe = np.asarray(x_ref, np.float64), engine["x"].astype(np.float64)
    return {"against": against, "rel_x": float(np.linalg.norm(xe - xr) / max(np.linalg.norm(xr), 1e-300)),
            "rel_optval": abs(engine["optval"] - optval_ref) / max(abs(optval_ref), 1e-300),
            "iterations_reference": int(it_ref), "iterations_engine": engine["iterations"] + 1, "tolerance": 1e-4}


def parity_from_fixture(cfg, fx, engine):
    """`parity_vs_reference` against the committed solutions of the compiled reference on exactly this
    (A, b, lambda) (tests/golden/make_c2_reference*.py, make_c3_reference.py, make_c4_reference.py)."""
    what = "tests/golden/%s: the compiled reference" % cfg["fixture"]
    if "x_fp64" in fx:   # dense fixtures hold both builds; the fp64 one is the algorithm without rounding noise
        par = _parity(engine, fx["x_fp64"], float(fx["optval_fp64"]), int(fx["iterations_fp64"]) + 1,
                      what + "'s fp64 build (PogsD) on exactly this (A, b, lambda), run to convergence in the build container")
        p32 = _parity(engine, fx["x"], float(fx["optval"]), int(fx["iterations"]) + 1, "")
        par["rel_x_vs_reference_fp32_build"] = p32["rel_x"]
        par["iterations_reference_fp32_build"] = p32["iterations_reference"]
        return par
    par = _parity(engine, fx["x"], float(fx["optval"]), int(fx["iterations"]) + 1,
                  what + " (PogsSparseS, fp32) on exactly this (A, b, lambda), run to convergence in the build container")
    # the reference adds the 2.5e6 terms of its optval in fp32 (prox_lib.h:521-529): 1e-3 off its own fp64 value
    par["note_optval"] = "the reference sums optval in fp32 over m + n terms; objective_at_x in the fixture is the fp64 recomputation"
    return par


# what the CPU leg runs per workload: (BLAS build of the compiled reference, max_iter; None = the default 2500, "cap" =
# --cpu-iters).  c2 with OpenBLAS gets 400 iterations: enough to converge where its fp32 build converges at all (154
# iterations in the build container), 20 s if it does not.
CPU_PLAN = {"c2": [("openblas", 400), ("mkl", "cap")], "c3": [("openblas", None)]}


def extrapolate_cpu(out, world, m, n):
    """N > 1: `out` was measured on rank 0's OWN shard (m x n, a whole problem of one GPU's size).  The job's
    problem has N m rows -- more elements than the reference can index at C5 (8e9: its vector and matrix views
    carry int sizes, src/cpu/include/gsl/gsl_vector.h:135-138, gsl_blas.h:34-37) -- and every per-iteration term of
    the reference (two gemv passes, the prox over m, the BLAS-1 algebra) is linear in m at fixed n, so its
    iteration time is taken as N x the shard's ("extrapolate t_iter ~ m from C2 and say
    so").  `value` becomes the whole problem's ADMM it/s; the measured rate stays as `value_on_one_shard`, which is
    ALSO the CPU's figure in the line's own unit (shard iterations per second summed over the job: N shards at
    1 / (N t_iter) each)."""
    v = out.get("value")
    if v is None or world <= 1:
        return out
    out["value_on_one_shard"] = v
    out["value"] = v / world
    out["unit"] = "it/s of the whole %d x %d problem" % (m * world, n)
    out["value_in_metric_unit"] = v
    out["extrapolated"] = ("t_iter ~ m (SURVEY.md section 8(d)): measured on rank 0's %d x %d shard, divided by N = %d; the "
                           "reference cannot index the whole problem's %.1e elements at C5 (gsl_vector.h:135-138, "
                           "gsl_blas.h:34-37)" % (m, n, world, float(m) * world * n))
    for k in ("time_to_converge_s", "init_s", "loop_s"):
        if k in out:
            out[k + "_on_one_shard"] = out.pop(k)
    out["sample"] = "rank 0's shard as a problem of its own (%s)" % out.get("sample", "")
    return out


def cpu_baseline(name, cfg, A_host, f, g, args, engine, world=1):
    """Times the reference CPU path on this box's host cores, on the SAME (A, f, g), WHOLE workload:
    no row sample, nothing scaled.  Returns (cpu_baseline dict, live parity dict or None).
    (world > 1: rank 0's own shard, one build only, extrapolate_cpu() applied -- see there.)

    Dense (c2, c3): the compiled reference (`kind` "reference"; a clean subprocess -- it must not share
    a process with torch), `value` = iterations / (Total - Init) from its own summary line
    (src/cpu/pogs.cpp:485-490).  Two builds of the same six sources (oracle/Makefile): against
    scipy's OpenBLAS, which threads its gemv -- SURVEY.md section 8(d)'s "best configuration" for the
    dense path, BLAS threads = granted cores -- and against MKL, the build the oracle is pinned to,
    whose sgemv runs on ONE thread on the GPU box's AMD host (profiles/r03_ref_cpu_diagnosis.md).
    The top-level fields are the best build's; `builds` holds each.  c2: the MKL build is capped at
    --cpu-iters ADMM iterations -- on that host it does not reach the default tolerances and runs twenty
    minutes into max_iter (--cpu-full lifts the caps); its first iterations are its cheapest (no
    exact-residual passes yet), so the cap flatters the CPU side -- the OpenBLAS build gets 400
    iterations, which is to convergence where it converges.  c3 runs to convergence.
    Sparse (c4): the OpenMP oracle port on the whole workload to convergence (`kind` "port": the
    reference's sparse path is single-threaded as built and needs 21 minutes,
    tests/golden/make_c4_reference.py), OpenMP threads = granted cores."""
    import numpy as np

    import oracle_binding as ob

    sparse = hasattr(A_host, "indptr")
    m, n = A_host.shape
    dt = np.float32
    soa = lambda fv: {k: getattr(fv, k) for k in "habcde"}  # noqa: E731
    fs, gs = soa(f), soa(g)
    out = {"unit": "it/s", "host_threads_visible": os.cpu_count() or 1, "cpu_quota": ob.cpu_quota()}
    if sparse:
        ob.oracle_set_threads()
        r = ob.oracle_solve(A_host, fs, gs, dtype=dt)
        t_init, t_loop = r["info"]["t_init"], r["info"]["t_loop"]
        iters = r["iterations"] + 1
        out.update(kind="port", cores=ob.cpu_quota(), value=iters / max(t_loop, 1e-9), time_to_converge_s=t_init + t_loop,
                   init_s=t_init, loop_s=t_loop, iterations=iters, converged=r["status"] == 0,
                   best={"build": "OpenMP oracle port", "threads": ob.cpu_quota(), "value": iters / max(t_loop, 1e-9)},
                   sample="whole workload %dx%d nnz %d fp32 to convergence, OpenMP oracle port (oracle/pogs_oracle.cpp, pinned to "
                          "the reference in tests/), %d threads: %d iterations, total %.1f s, init %.1f s"
                          % (m, n, A_host.nnz, ob.cpu_quota(), iters, t_init + t_loop, t_init))
        if world > 1:
            return extrapolate_cpu(out, world, m, n), None
        return out, _parity(engine, r["x"], r["optval"], iters, "oracle port, same A, b, lambda, default tolerances, whole workload, this run")
    builds, live = {}, None
    plan = CPU_PLAN.get(name, [("openblas", "cap")])
    if world > 1 or args.cpu_best_only:
        plan = plan[:1]   # the best build only: the other ranks are waiting
    for blas, cap in plan:
        if not ob.ref_available(blas):
            builds[blas] = {"value": None, "sample": "oracle/_ref build for %s missing" % blas}
            continue
        cap_it = None if (cap is None or args.cpu_full) else (args.cpu_iters if cap == "cap" else int(cap))
        t_start = time.time()
        try:
            r = ob.ref_solve(A_host, fs, gs, dtype=dt, verbose=1, timeout=args.cpu_budget_s, blas=blas,
                             **({"max_iter": cap_it} if cap_it else {}))
        except Exception as e:
            builds[blas] = {"value": None, "sample": "failed: %r" % (e,)}
            continue
        t_total, t_init = r.get("t_total", r["wall_s"]), r.get("t_init", 0.0)
        iters = r["iterations"] + 1
        converged = r["status"] == 0
        lib = "oracle/_ref/libpogs_cpu_openblas.so (scipy OpenBLAS)" if blas == "openblas" else "oracle/_ref/libpogs_cpu.so (MKL)"
        d = dict(value=iters / max(t_total - t_init, 1e-9), init_s=t_init, loop_s=t_total - t_init, iterations=iters,
                 converged=converged, threads=ob.ref_threads(),
                 sample="whole workload %dx%d fp32, compiled reference %s, %s: %d iterations%s, Total %.1f s, Init %.1f s "
                        "(its own summary line); elapsed with process start and input hand-over %.1f s"
                        % (m, n, lib, "to convergence" if converged else "max_iter = %d" % (cap_it or 2500), iters,
                           "" if converged else " (not converged: capped, see cpu_baseline() in bench.py)", t_total, t_init,
                           time.time() - t_start))
        if converged:
            d["time_to_converge_s"] = t_total
            if live is None:
                live = _parity(engine, r["x"], r["optval"], iters, "compiled reference (%s), same A, b, lambda, default "
                                                                   "tolerances, whole workload, this run" % lib)
        builds[blas] = d
    ok = {k: v for k, v in builds.items() if v.get("value")}
    if not ok:
        raise RuntimeError("no reference build ran: %r" % (builds,))
    best = max(ok, key=lambda k: ok[k]["value"])
    out.update(kind="reference", cores=ob.ref_threads(), threads_env=ob.ref_env_note() + " OPENBLAS_NUM_THREADS=%d" % ob.ref_threads(),
               builds=builds, best={"build": best, "threads": ok[best]["threads"], "value": ok[best]["value"]})
    out.update({k: ok[best][k] for k in ("value", "init_s", "loop_s", "iterations", "converged", "sample")})
    if "time_to_converge_s" in ok[best]:
        out["time_to_converge_s"] = ok[best]["time_to_converge_s"]
    if world > 1:
        return extrapolate_cpu(out, world, m, n), None   # (the shard alone is not the sharded problem: no live parity)
    return out, live


def pick_windows(K, L, per_step_s, min_s=0.25, max_windows=40):
    """Number of back-to-back windows of K steps for a workload whose solve takes L iterations: W K as close to a
    whole number k of solves as W <= max_windows allows, the smallest k that makes the timed stretch at least `min_s`
    seconds.  Returns (W, fraction of a solve by which W K misses k L)."""
    if K >= 3 * L:
        return 1, abs(K - round(K / L) * L) / float(K)
    best = None
    for k in range(1, 33):
        w = max(1, int(round(k * L / float(K))))
        if w > max_windows:
            break
        err = abs(w * K - k * L) / float(L)
        long_enough = w * K * per_step_s >= min_s
        # (a stretch that stays too short whatever k: the longest one that is still close to whole solves)
        cand = (0, round(err, 3), w) if long_enough else (1, 0 if err <= 0.06 else 1, -w)
        if best is None or cand < best[0]:
            best = (cand, w, err)
        if long_enough and err <= 0.06:
            break
    if best is None:
        return max(1, min(max_windows, int(round(L / float(K))))), 1.0
    return best[1], best[2]


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


def live_traffic(name, kernel_substr, budget_s=150.0, device_index=None):
    """HBM bytes per launch of the dominant kernel MEASURED NOW: this script again on the same
    workload (a few steps) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes,
    kernel trace only), summarised exactly as scripts/pmc_summary.py does -- FETCH_SIZE / WRITE_SIZE are
    KiB, the read side doubled (guides/MI355X_MICROARCH.md: gfx950 reports half of a wide coalesced
    streaming read), launches of the device-resident CG loop that returned at once left out.
    Returns (bytes per launch, description) or (None, reason)."""
    import csv
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this run is itself being profiled"
    tmp = tempfile.mkdtemp(prefix="pogs_pmc_", dir="/tmp")
    t_start = time.time()
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "g", "--",
                   sys.executable, os.path.abspath(__file__), "--config", name, "--steps", "10", "--warmup", "2",
                   "--traffic-child", "--no-cpu-baseline", "--no-secondary", "--no-live-traffic"]
            left = budget_s - (time.time() - t_start)
            if left < 20:
                return None, "time: budget of the counter passes used up"
            # the pass is a ONE-GPU run of the workload (at N > 1: of one rank's shard size -- the kernel, its
            # arguments and its traffic are the same on every rank): nothing of this job's launcher, rendezvous or
            # forced-communicator environment may reach it
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
                                "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "POGS_AMD_FORCE_DIST",
                                "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
                                "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_ERROR_FILE", "OMP_NUM_THREADS")
                   and not k.startswith("TORCHELASTIC_")}
            env["TMPDIR"] = "/tmp"
            if device_index is not None:
                env["HIP_VISIBLE_DEVICES"] = str(device_index)   # rank 0's GPU (the others are idle, waiting)
            # its own process group, so that a pass that hangs can be ended together with whatever it started
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                  start_new_session=True)
            try:
                rc = pr.wait(timeout=left)
            except subprocess.TimeoutExpired:
                import signal

                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return None, "time: the %s pass ran into its budget" % counter
            if rc != 0:
                return None, "rocprofv3 --pmc %s pass failed (rc %d)" % (counter, rc)
            per_launch = []
            for root, _, files in os.walk(out):
                for fn in files:
                    if not fn.endswith("counter_collection.csv"):
                        continue
                    with open(os.path.join(root, fn)) as fh:
                        for row in csv.DictReader(fh):
                            if row["Counter_Name"] != counter:
                                continue
                            kn = re.sub(r"pogs_amd::|\(anonymous namespace\)::", "", row["Kernel_Name"])
                            if kernel_substr in kn:
                                per_launch.append(float(row["Counter_Value"]))
            if not per_launch:
                return None, "no launch of the kernel in the %s pass" % counter
            vals[counter] = per_launch
        f, w = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
        if len(f) == len(w) and max(f) > 0:
            keep = [i for i in range(len(f)) if f[i] >= 0.02 * max(f)]
            f, w = [f[i] for i in keep], [w[i] for i in keep]
        fe, wr = sum(f) / len(f) * 1024.0, sum(w) / len(w) * 1024.0
        return 2.0 * fe + wr, ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (kernel trace only) of "
                               "`bench.py --config %s --steps 10 --warmup 2` spawned after the timed part, %d launches of the "
                               "kernel, read side x 2 (the gfx950 half count of wide streaming reads), %.0f s for both passes"
                               % (name, len(f), time.time() - t_start))
    except Exception as e:   # a profiler hiccup must not cost the bench line
        return None, "counter pass: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def unsharded_parity(cfg, m, n, world, dev, local, res, shard_b_sums):
    """Rank 0, after the timed region of an N > 1 run: regenerates every rank's rows (torch_rows with the
    ranks' seeds, on this device), checks them against the sums of b the ranks reported, solves the whole
    (m world) x n problem UNSHARDED on this GPU and holds the row-sharded result `res` against it."""
    import numpy as np
    import torch

    import pogs_amd
    from pogs_amd import graph as G

    from pogs_amd import _lib as L

    np_dtype = np.float64 if cfg["dtype"] == "f64" else np.float32
    # the library's pool may be sitting on tens of GB of idle blocks (the sharded handle was just closed): give them
    # back before torch asks the runtime for the whole matrix
    L.pool_trim(local)
    A_all = torch.empty((m * world, n), dtype=torch.float64 if cfg["dtype"] == "f64" else torch.float32, device=dev)
    b_parts = []
    for r in range(world):
        Ar, br = torch_rows(cfg, m, n, r, dev)
        # the regenerated rows are the rank's own
        assert abs(float(br.sum()) - shard_b_sums[r]) <= 1e-9 * max(1.0, abs(shard_b_sums[r])), "rank %d's rows differ" % r
        A_all[r * m:(r + 1) * m] = Ar
        b_parts.append(br)
        del Ar
    f_all, g_all = functions(cfg, G, np.concatenate(b_parts), n)
    with pogs_amd.Solver(A_all.data_ptr(), dtype=np_dtype, shape=(m * world, n), device_ptr=True, device=local) as s1:
        r1 = s1.solve(f_all, g_all)
        st1 = s1.stats()
    del A_all
    torch.cuda.empty_cache()
    par = _parity(res, r1["x"], r1["optval"], r1["iterations"] + 1,
                  "the same %d x %d problem solved UNSHARDED on rank 0's GPU by the engine (which is pinned to the compiled "
                  "reference at N = 1): the row-sharded RCCL solve must land on it" % (m * world, n))
    par["unsharded_it_per_s"] = (r1["iterations"] + 1) / max(st1["t_loop_s"], 1e-9)
    return par


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
        # Rehearsal of the N > 1 launcher path on a box with fewer GPUs than ranks (tests/test_gpu_bench.py;
        # POGS_AMD_BENCH_REHEARSAL=1, test-only): the ranks share the devices there are, the process group is gloo
        # and the solver handles are joined by the shared-memory communicator of the test transport plug-in
        # (tests/transport/test_transport.hip) -- RCCL refuses two ranks on one device.  Everything else is the
        # real run's code: rank environment, the 128-byte id broadcast, per-rank shards, the barriers, rank 0's
        # CPU leg and unsharded parity solve while the others wait, the line's assembly.
        self.rehearsal = os.environ.get("POGS_AMD_BENCH_REHEARSAL", "0") == "1" and self.world > 1
        if self.rehearsal:
            self.local = self.local % max(torch.cuda.device_count(), 1)
            os.environ.setdefault("POGS_AMD_TRANSPORT_PLUGIN",
                                  os.path.join(ROOT, "tests", "transport", "libpogs_test_transport.so"))
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        self.coll_dev = torch.device("cpu") if self.rehearsal else self.dev   # where the process group's tensors live
        self.dist = None
        self.force_dist = os.environ.get("POGS_AMD_FORCE_DIST", "0") == "1"  # exercise the RCCL path with 1 rank
        if self.world > 1 or self.force_dist:
            import torch.distributed as dist

            import datetime

            if self.force_dist and "MASTER_ADDR" not in os.environ:
                with socket.socket() as sk:   # a free port: the counter passes' children and other tests may run beside us
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
            # rank 0 works alone for minutes after the timed region (the CPU baseline, the unsharded parity solve,
            # the counter passes): no collective may time out meanwhile
            dist.init_process_group("gloo" if self.rehearsal else "nccl", timeout=datetime.timedelta(minutes=60))
            self.dist = dist
            # the other ranks wait for rank 0 on sockets (gloo), not spinning on a GPU collective: rank 0's CPU
            # baseline needs the host cores
            try:
                self.wait_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=60))
            except Exception:
                self.wait_group = None

    wait_group = None

    def barrier(self):
        import torch

        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def long_barrier(self):
        """Ranks meet after a stretch in which rank 0 worked alone (host-side wait where gloo is there)."""
        import torch

        torch.cuda.synchronize()
        if self.dist is not None:
            if self.wait_group is not None:
                self.dist.barrier(group=self.wait_group)
            else:
                self.dist.barrier()


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
        uid = torch.zeros(128, dtype=torch.uint8, device=env.coll_dev)
        if rank == 0:
            if env.rehearsal:   # an id of the test plug-in's shared-memory communicator instead of RCCL's
                env.shm_ids = getattr(env, "shm_ids", 0) + 1
                raw = ("POGSSHM:%d-%d-%s" % (os.getpid(), env.shm_ids, os.urandom(4).hex())).encode().ljust(128, b"\0")
            else:
                raw = pogs_amd.dist_unique_id()
            uid = torch.tensor(list(raw), dtype=torch.uint8, device=env.coll_dev)
        dist.broadcast(uid, 0)
        return (rank, world, m * world, bytes(uid.cpu().tolist()))

    np_dtype = np.float64 if cfg["dtype"] == "f64" else np.float32
    esize = np.dtype(np_dtype).itemsize
    A, b, A_host, fixture = make_problem(cfg, m, n, rank, dev, world)
    torch.cuda.synchronize()

    if sparse:
        # the CSR arrays resident in HBM, like the dense matrices: the timed setup starts from there
        csr_dev = (torch.from_numpy(np.ascontiguousarray(A.data, np_dtype)).to(dev),
                   torch.from_numpy(np.ascontiguousarray(A.indptr, np.int32)).to(dev),
                   torch.from_numpy(np.ascontiguousarray(A.indices, np.int32)).to(dev))
        torch.cuda.synchronize()

    def create():
        dist_arg = new_dist_arg()
        if sparse:
            return pogs_amd.Solver((csr_dev[0].data_ptr(), csr_dev[1].data_ptr(), csr_dev[2].data_ptr(), A.nnz),
                                   dtype=np_dtype, shape=(m, n), device_ptr=True, device=local, profile=PROFILE_EVERY,
                                   dist=dist_arg)
        from pogs_amd import _lib as L

        return pogs_amd.Solver(A.data_ptr(), dtype=np_dtype, shape=(m, n), device_ptr=True, device=local,
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
    # exactly K steps between barriers, in as many back-to-back windows as make the timed stretch a whole number of
    # SOLVES (pick_windows): the iterations of a solve do not cost the same (a CGLS projection takes 4 steps early in
    # a C4 solve, 1 in its middle, 2-3 plus the exact-residual products at its end: 1.5 / 0.53 / 1.06 ms per iteration),
    # and the metric is a solve's iterations over its loop time -- a window that covers the cheap
    # middle of a solve would flatter it.  The iterations run periodically (a converged solve restarts from the cold
    # start), so any stretch of k whole periods is unbiased whatever its phase.
    per_step_guess = max(st_solve["t_loop_s"] / max(st_solve["iterations"], 1), 1e-6)
    windows, cover = pick_windows(args.steps, int(res["iterations"]) + 1, per_step_guess)
    if dist is not None:   # every rank must run the same number of windows (barriers): rank 0's choice
        wt = torch.tensor([windows], dtype=torch.int64, device=env.coll_dev)
        dist.broadcast(wt, 0)
        windows = int(wt.item())
    times = []
    solver.reset_stats()
    for _ in range(windows):
        env.barrier()
        t0 = time.time()
        solver.iterate(args.steps)
        env.barrier()
        elapsed = time.time() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=env.coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        times.append(elapsed)
    # the windows are consecutive stretches of the same real solves: their MEAN is the iteration rate
    # (the median would drop the windows that hold a solve's dearer iterations -- a missed rho
    # prediction's extra pass, the two-step CG projections late in a sparse solve)
    elapsed = sum(times) / len(times)
    st = solver.stats()
    nranks_comm = st.get("comm_nranks", 0)   # as ncclCommCount reports it (0: no communicator)
    solver.close()

    # create / solve / destroy, HANDLE_CYCLES times: what a caller of the one-shot ABI pays per call
    # (the reference builds and destroys its solver inside PogsD/PogsS, src/interface_c/pogs_c.cpp:19-20);
    # the library's device pool (pogs_amd/csrc/common.h) is what keeps every cycle at the speed of the fastest
    cycles = None
    if world == 1 and dist is None and not args.traffic_child:
        from pogs_amd import _lib as L

        p0 = L.pool_stats(local)
        ci, ct, cc = [], [], []
        for _ in range(HANDLE_CYCLES):
            torch.cuda.synchronize()
            t0 = time.time()
            s_ = create()
            t1 = time.time()
            r_ = s_.solve(f, g)
            t2 = time.time()
            s_.close()
            t3 = time.time()
            assert r_["iterations"] == res["iterations"] and r_["status"] == res["status"]
            ci.append(t1 - t0)
            ct.append(t2 - t0)
            cc.append(t3 - t0)
        p1 = L.pool_stats(local)
        cycles = {"n": HANDLE_CYCLES, "init_s": ci, "time_to_converge_s": ct, "create_solve_destroy_s": cc,
                  "max_init_s": max(ci), "max_time_to_converge_s": max(ct),
                  "pool": {"hipMalloc_calls": p1["mallocs"] - p0["mallocs"], "hipFree_calls": p1["frees"] - p0["frees"],
                           "blocks_reused": p1["reuses"] - p0["reuses"], "cached_bytes": p1["cached_bytes"]}}

    # The same create + solve with the setup's two default-on shortcuts switched off (the fp32 Gram product on the
    # native fp32 MFMA instead of the two-way fp16 split, all 50 Sinkhorn-Knopp passes): what the default saves, and
    # what a caller who wants the reference's setup arithmetic to the last bit pays.
    exact_setup = None
    if not sparse and cfg["dtype"] == "f32" and world == 1 and dist is None and not args.traffic_child \
            and args.projector == "default":
        saved = {k: os.environ.get(k) for k in ("POGS_AMD_GRAM", "POGS_AMD_SK_FULL")}
        os.environ.update(POGS_AMD_GRAM="fp32", POGS_AMD_SK_FULL="1")   # read at handle creation
        try:
            torch.cuda.synchronize()
            t0 = time.time()
            s_ = create()
            t1 = time.time()
            r_ = s_.solve(f, g)
            t2 = time.time()
            st_ = s_.stats()
            s_.close()
            exact_setup = {"time_to_converge_s": t2 - t0, "init_s": t1 - t0, "iterations": r_["iterations"] + 1,
                           "status": r_["status"], "gram_ms": st_["gram_ms"], "equil_ms": st_["equil_ms"],
                           "rel_x_vs_default_setup": float(np.linalg.norm(r_["x"].astype(np.float64) - res["x"].astype(np.float64))
                                                           / max(np.linalg.norm(res["x"].astype(np.float64)), 1e-300)),
                           "switches": "POGS_AMD_GRAM=fp32 POGS_AMD_SK_FULL=1"}
        except Exception as e:
            exact_setup = {"error": repr(e)[:300]}
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    # What a caller of the reference's one-shot entry point sees (src/interface_c/pogs_c.cpp:19-52: PogsS / PogsD with
    # a HOST matrix; the totals it prints, src/cpu/pogs.cpp:485-490): the whole call with the upload of A inside it.
    # The upload is reported on its own ("H2D separately") from a handle created on the
    # same host buffer, whose t_h2d_s is the copy alone.  Never part of `value`.
    one_shot = None
    if not sparse and A_host is not None and world == 1 and dist is None and not args.traffic_child \
            and args.projector == "default" and (m, n) == (cfg["m"], cfg["n"]):
        try:
            torch.cuda.synchronize()
            t0 = time.time()
            r_ = G._solve_graph_form(A_host, f, g, dtype=np_dtype)
            t_call = time.time() - t0
            t0 = time.time()
            with pogs_amd.Solver(A_host, dtype=np_dtype, device=local) as s_:
                t_create = time.time() - t0
                st_ = s_.stats()
            one_shot = {"one_shot_host_call_s": t_call, "h2d_s": st_["t_h2d_s"], "create_from_host_s": t_create,
                        "iterations": r_["iterations"] + 1, "status": r_["status"],
                        "h2d_gb_per_s": float(m) * n * esize / max(st_["t_h2d_s"], 1e-9) / 1e9,
                        "what": "%s(ROW_MAJ, host A, ...) through the C ABI, wall clock around the call: upload of A (pageable "
                                "host memory) + setup + loop + results back" % ("PogsD" if cfg["dtype"] == "f64" else "PogsS")}
            assert r_["iterations"] == res["iterations"] and r_["status"] == res["status"]
        except Exception as e:
            one_shot = {"error": repr(e)[:300]}

    # N > 1: the shards are row ranges of ONE problem; rank 0 regenerates it whole and, when it fits,
    # solves it unsharded -- the sharded solution must land on that solve's
    unsharded = None
    if world > 1 and not sparse and args.projector == "default":
        bsum = torch.tensor([float(np.asarray(b, np.float64).sum())], dtype=torch.float64, device=env.coll_dev)
        sums = [torch.zeros_like(bsum) for _ in range(world)]
        dist.all_gather(sums, bsum)
        if rank == 0 and float(m) * world * n * esize <= UNSHARDED_CHECK_MAX_BYTES:
            try:
                unsharded = unsharded_parity(cfg, m, n, world, dev, local, res, [float(v.item()) for v in sums])
            except Exception as e:   # never take the line down
                unsharded = {"error": repr(e)[:300]}
        env.long_barrier()

    line = None
    if rank == 0 and getattr(env, "peak_measured", None) is None and not args.traffic_child:
        # the measured read ceiling of THIS device next to the data-sheet peak ("state
        # both"): the library's read-bandwidth probe on 4 GB, nothing else running on the GPU
        try:
            from pogs_amd import _lib as L

            torch.cuda.synchronize()
            gbs, pat = L.read_bandwidth(local, 4 << 30, 10)
            env.peak_measured = {"gb_per_s": gbs, "pattern": pat}
        except Exception as e:
            env.peak_measured = {"gb_per_s": None, "error": repr(e)[:200]}
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
            # (fp32 logistic solves on rows of 1025 .. 1280 float4 vectors run the prefetching form, stream.h)
            pf = cfg["kind"] != "dense_lasso" and cfg["dtype"] != "f64" and 1024 < (n + 3) // 4 <= 1280 and m > n
            kernel = "%s<FusedIterOp> (the one pass over A per iteration)" % ("stream_rows2_pf_kernel" if pf else "stream_rows2_kernel")
            kernel_key = "stream_rows2_"   # both forms (a solve launches one of them; its element type is the config's)
            one_pass = esize * (m * n + 0.5 * n * n)  # A once + the lower triangle of W = L^-1
            two_pass = esize * (2.0 * m * n + n * n)  # the reference algorithm
            iteration = {"bytes_model": "one-pass engine: A once + the lower triangle of W per iteration",
                         "bytes": one_pass, "frac": one_pass * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                         "passes_over_A_per_iteration": st["matvecs"] / max(steps_total, 1),
                         "vs_reference_algorithm_bytes": {
                             "bytes": two_pass, "note": "the reference reads A twice per iteration (+2 on exact-residual "
                                                        "iterations); this ratio is a speed-up over that byte model, NOT a "
                                                        "roofline fraction",
                             "ratio_to_hbm_peak": two_pass * args.steps / elapsed / 1e9 / HBM_PEAK_GBS}}
            wname = "solve_lasso" if cfg["kind"] == "dense_lasso" else "solve_logistic"
            workload = ("%s dense %s A=%dx%d per GPU, lambda=%g, default tolerances (BASELINE.json configs[%d]%s%s)"
                        % (wname, "fp64" if cfg["dtype"] == "f64" else "fp32", m, n, cfg["lambd"], cfg["cfg_index"],
                           " with the matrix widened to the reference Python layer's dtype, python/pogs/graph.py:281-288"
                           if cfg["dtype"] == "f64" else "",
                           "" if world == 1 else "; row-sharded %dx%d" % (m * world, n)))
            projector = "direct (MFMA Gram + Cholesky)"
            if args.projector == "cgls":
                passes = st["matvecs"] / max(steps_total, 1)
                kernel = "stream_rows_kernel (every pass over A of the loop: A p, A^T r, A x of CGLS and the residual passes)"
                kernel_key = "stream_rows_kernel<%s" % ("double" if cfg["dtype"] == "f64" else "float")
                iteration = {"bytes_model": "passes over A per iteration x s m n bytes (matrix-free CGLS projector)",
                             "passes_over_A_per_iteration": passes, "cg_per_iteration": st["cg_iters"] / max(steps_total, 1),
                             "bytes": passes * esize * m * n, "frac": passes * esize * m * n * args.steps / elapsed / 1e9 / HBM_PEAK_GBS}
                projector = "CGLS on the dense matrix (matrix-free)"
                workload += " [--projector cgls]"
        traffic, traffic_src, traffic_fresh = (pmc_traffic(name, kernel_key)
                                               if (m, n) == (cfg["m"], cfg["n"]) and args.projector == "default"
                                               else (None, None, None))
        line = {
            "metric": "admm_iterations_per_sec_%s_%s (per-GPU shard, summed over GPUs)"
                      % (cfg["kind"], "fp64" if cfg["dtype"] == "f64" else "fp32"),
            "value": its, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
            "windows": windows, "window_s": times,
            "windows_cover": "%d x %d = %d iterations = %.2f solves of %d" % (windows, args.steps, windows * args.steps,
                                                                              windows * args.steps / float(res["iterations"] + 1),
                                                                              res["iterations"] + 1),
            # (the scalars a reader of the driver's record needs sit in `config` and `roofline`, which it keeps whole:
            # wall-clock-to-converge is half of BASELINE.json's metric)
            "config": {"workload": workload, "name": name, "rows_per_gpu": m, "cols": n, "projector": projector,
                       "parallelism": "row-shard x%d" % world, "rccl_nranks": nranks_comm,
                       "time_to_converge_s": init_s + solve_s, "init_s": init_s, "loop_s": st_solve["t_loop_s"],
                       "solve_iterations": res["iterations"] + 1, "solve_status": res["status"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "peak_datasheet": HBM_PEAK_GBS,
                         "peak_measured": (getattr(env, "peak_measured", None) or {}).get("gb_per_s"),
                         "frac_of_peak_measured": (achieved / env.peak_measured["gb_per_s"]
                                                   if (getattr(env, "peak_measured", None) or {}).get("gb_per_s") else None),
                         "peak_measured_how": "PogsAmdReadBandwidth: best of two read-only kernels over 4 GB on this device, "
                                              "10 launches each (pattern: %s)" % (getattr(env, "peak_measured", None) or {}).get("pattern"),
                         "iteration_frac": iteration["frac"],
                         "traffic_source": ("static: %s (rocprofv3 --pmc passes of this command, committed; not "
                                            "measured in this run; %s)"
                                            % (traffic_src, "collected on the kernel sources of this build" if traffic_fresh
                                               else "STALE: the kernel sources have changed since the counters were collected"
                                               if traffic_fresh is False else "no source hash in the file")) if traffic_src else None,
                         "kernel": kernel, "counter_kernel_match": kernel_key, "bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": avg_ms, "launches": st["stream_launches"],
                         "launch_sampling": "HIP events around every %d-th launch of the kernel in the timed region" % PROFILE_EVERY,
                         "iteration": iteration},
            "time_to_converge_s": init_s + solve_s, "init_s": init_s, "loop_s": st_solve["t_loop_s"],
            "first_handle_of_the_process": {"init_s": init_cold_s, "time_to_converge_s": init_cold_s + solve_s,
                                            "note": "the process's first solver handle also pays for HIP stream creation and "
                                                    "code-object loading (and takes its device memory from the HIP runtime, not from "
                                                    "the library's pool); init_s / time_to_converge_s are a second handle's"},
            "time_to_converge_includes": "the one-time setup with its two default-on shortcuts -- the fp32 Gram product as a "
                                         "two-way fp16 split on the matrix cores (POGS_AMD_GRAM=fp32: native fp32 MFMA) and "
                                         "Sinkhorn-Knopp's common-factor tail in closed form (POGS_AMD_SK_FULL=1: all 50 passes); "
                                         "both are held against the exact path at full size in tests/test_gpu_fullsize.py",
            "solve_iterations": res["iterations"] + 1, "solve_status": res["status"],
            "exact_residual_iters": st_solve["exact_iters"],
            "setup_ms": {k: st_solve[k] for k in ("equil_ms", "normest_ms", "gram_ms", "chol_ms", "trtri_ms")},
        }
        if not sparse:
            line["gram_tflops"] = st_solve["gram_flops"] / max(st_solve["gram_ms"], 1e-9) / 1e9
        if env.rehearsal:
            line["config"]["rehearsal"] = ("POGS_AMD_BENCH_REHEARSAL=1: %d ranks on %d GPU(s), process group gloo, solver handles joined by "
                                           "the test plug-in's shared-memory communicator -- a plumbing rehearsal, not a measurement"
                                           % (world, torch.cuda.device_count()))
        if cycles is not None:
            line["handle_cycles"] = cycles
            line["config"]["handle_cycles_max_time_to_converge_s"] = cycles["max_time_to_converge_s"]
            line["config"]["handle_cycles_max_init_s"] = cycles["max_init_s"]
        if exact_setup is not None:
            line["exact_setup"] = exact_setup
            line["time_to_converge_exact_setup_s"] = exact_setup.get("time_to_converge_s")
            line["config"]["time_to_converge_exact_setup_s"] = exact_setup.get("time_to_converge_s")
        if one_shot is not None:
            line["one_shot_host_call"] = one_shot
            line["config"]["one_shot_host_call_s"] = one_shot.get("one_shot_host_call_s")
            line["config"]["h2d_s"] = one_shot.get("h2d_s")
        if fixture is not None:
            try:
                line["parity_vs_reference"] = parity_from_fixture(cfg, fixture, res)
            except Exception as e:
                line["parity_vs_reference"] = {"error": repr(e)[:300]}
        elif unsharded is not None:
            line["parity_vs_reference"] = unsharded
    if rank == 0 and isinstance(line.get("parity_vs_reference"), dict) and "rel_x" in line["parity_vs_reference"]:
        par = line["parity_vs_reference"]
        line["config"]["parity_rel_x"] = par["rel_x"]
        line["config"]["parity_iterations"] = "%d engine / %d reference" % (par["iterations_engine"], par["iterations_reference"])
    if rank == 0 and with_cpu and cfg["dtype"] == "f32":
        try:
            if sparse:
                A_host = A
            elif A_host is None:
                A_host = A.cpu().numpy()
            del A   # the GPU copy is not needed any more; the reference gets the host copy
            torch.cuda.empty_cache()
            line["cpu_baseline"], live = cpu_baseline(name, cfg, A_host, f, g, args, res, world)
            if live is not None and "parity_vs_reference" in line:
                line["parity_vs_reference"]["live_cpu_run"] = live
            elif live is not None:
                line["parity_vs_reference"] = live
        except Exception as e:  # the baseline must never take the bench line down
            line["cpu_baseline"] = {"value": None, "unit": "it/s", "cores": os.cpu_count(), "kind": "none",
                                    "sample": "failed: %r" % (e,)}
    A = None
    if sparse:
        csr_dev = None
    torch.cuda.empty_cache()
    try:   # the next workload's matrices come from torch: the pool's idle blocks of this one go back to the runtime
        from pogs_amd import _lib as L

        L.pool_trim(local)
    except Exception:
        pass
    return line


SUMMARY_MAX_CHARS = 6000   # the driver keeps an 8 KB tail of stdout and parses its last line (BENCH_r05: a 24 KB line was lost)
DETAIL_PREFIX = "BENCH_DETAIL "


def _short(x, sig=6):
    """Floats to `sig` significant digits, strings cut to 160 characters, containers walked."""
    if isinstance(x, bool) or x is None or isinstance(x, int):
        return x
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, str):
        return x if len(x) <= 160 else x[:157] + "..."
    if isinstance(x, dict):
        return {k: _short(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _workload_summary(d):
    """The few figures of one secondary workload the contract line keeps (its whole dictionary is in the detail line)."""
    if not isinstance(d, dict) or not d.get("value"):
        return _short(_pick(d, ("value", "error")))
    rf, par, cb = d.get("roofline") or {}, d.get("parity_vs_reference") or {}, d.get("cpu_baseline") or {}
    out = _pick(d, ("value", "unit", "dtype", "ms_per_step", "time_to_converge_s", "init_s", "solve_iterations", "solve_status"))
    out["workload"] = (d.get("config") or {}).get("workload", "")[:110]
    out["roofline"] = _pick(rf, ("frac", "achieved", "traffic", "bytes_per_launch", "avg_launch_ms", "iteration_frac",
                                 "frac_of_peak_measured"))
    out["roofline"]["kernel"] = (rf.get("kernel") or "")[:60]
    out["parity_rel_x"] = par.get("rel_x")
    if "iterations_engine" in par:
        out["parity_iterations"] = "%d engine / %d reference" % (par["iterations_engine"], par["iterations_reference"])
    if cb:
        out["cpu_baseline"] = _pick(cb, ("value", "cores", "kind", "time_to_converge_s", "converged"))
    return _short(out)


def summary_line(line):
    """The contract line: starts with `metric`, carries `roofline` and `cpu_baseline` whole enough to check, the
    headline workload's wall-clock and parity scalars, and a short summary per secondary workload; at most
    SUMMARY_MAX_CHARS characters (the reference prints its totals in two short lines, src/cpu/pogs.cpp:485-490)."""
    rf, cf = line["roofline"], line["config"]
    out = _pick(line, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    out["config"] = _pick(cf, ("workload", "name", "rows_per_gpu", "cols", "projector", "parallelism", "rccl_nranks", "rehearsal"))
    out["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "bytes_per_launch",
                                 "avg_launch_ms", "launches", "peak_measured", "frac_of_peak_measured", "iteration_frac"))
    src = rf.get("traffic_source") or rf.get("traffic_live")
    if src:
        out["roofline"]["traffic_source"] = src[:100]
    if isinstance(rf.get("iteration"), dict):
        out["roofline"]["iteration_bytes"] = rf["iteration"].get("bytes")
    cb = line.get("cpu_baseline")
    if cb is not None:
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "time_to_converge_s", "init_s", "loop_s",
                                         "iterations", "converged", "extrapolated", "host_threads_visible"))
    out.update(_pick(line, ("time_to_converge_s", "init_s", "loop_s", "solve_iterations", "solve_status",
                            "time_to_converge_exact_setup_s", "setup_ms", "gram_tflops", "windows", "windows_cover")))
    if "handle_cycles" in line:
        out["handle_cycles_max_time_to_converge_s"] = line["handle_cycles"]["max_time_to_converge_s"]
        out["handle_cycles_max_init_s"] = line["handle_cycles"]["max_init_s"]
    if isinstance(line.get("one_shot_host_call"), dict):
        out["one_shot_host_call_s"] = line["one_shot_host_call"].get("one_shot_host_call_s")
        out["h2d_s"] = line["one_shot_host_call"].get("h2d_s")
    par = line.get("parity_vs_reference")
    if isinstance(par, dict):
        out["parity"] = _pick(par, ("rel_x", "rel_optval", "iterations_engine", "iterations_reference", "tolerance",
                                    "rel_x_vs_reference_fp32_build", "against", "error"))
    if "secondary" in line:
        out["secondary"] = {k: _workload_summary(v) for k, v in line["secondary"].items()}
    out["detail"] = "stdout line before this one (prefix %r) and gpurun_out/bench_detail.json" % DETAIL_PREFIX.strip()
    out = _short(out)
    # a hard stop, never reached by the shapes above (tests/test_gpu_bench.py holds the line under the limit)
    for drop in ("detail", "setup_ms", "windows_cover", "handle_cycles_max_init_s", "h2d_s"):
        if len(json.dumps(out)) <= SUMMARY_MAX_CHARS:
            break
        out.pop(drop, None)
    return out


def write_detail(detail_line):
    """The long record: one prefixed stdout line BEFORE the contract line, and a file next to the profiles."""
    if detail_line is None:
        return
    print(DETAIL_PREFIX + detail_line, flush=True)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_detail.json"), "w") as fh:
            fh.write(detail_line + "\n")
    except OSError:
        pass


LIVE_TRAFFIC_LATEST_START_S = 420.0   # the counter passes are skipped when the run has already taken longer (a cold box)


def main():
    t_main = time.time()
    args = parse()
    maybe_spawn(args)
    import oracle_binding as ob

    # host threads (OpenMP of the oracle port, torch's CPU ops): what the container may really use
    os.environ.setdefault("OMP_NUM_THREADS", str(ob.cpu_quota()))
    env = Env(args)
    head = args.config or "c2"
    # N > 1: rank 0 times the reference on its own shard after the timed region (cpu_baseline / extrapolate_cpu)
    line = run_config(env, head, with_cpu=not args.no_cpu_baseline)
    # the driver's invocation (no --config, one GPU): c3, c4 (with their CPU legs) and c2 in fp64 under the same clock
    if args.config is None and env.world == 1 and not args.no_secondary and not (args.m or args.n) \
            and args.projector == "default":
        keep = ("metric", "value", "unit", "dtype", "ms_per_step", "steps", "windows", "window_s", "windows_cover", "config", "roofline",
                "time_to_converge_s", "init_s", "handle_cycles", "solve_iterations", "solve_status", "setup_ms", "gram_tflops",
                "parity_vs_reference", "cpu_baseline", "exact_setup", "time_to_converge_exact_setup_s", "one_shot_host_call")
        sec = {}
        for name in ("c3", "c4", "c2f64"):
            try:
                d = run_config(env, name, with_cpu=not args.no_cpu_baseline)
                sec[name] = {k: d[k] for k in keep if k in d}
            except Exception as e:
                sec[name] = {"value": None, "error": repr(e)[:300]}
        if line is not None:
            line["secondary"] = sec
    # roofline.traffic of the headline workload from counters collected NOW (everything timed is done; the
    # committed summary stays in the line as `traffic_static` for comparison)
    # (N > 1: rank 0 alone, on its own GPU, with the shard's size -- the other ranks wait in long_barrier below)
    if line is not None and not args.no_live_traffic and not args.traffic_child \
            and not (args.m or args.n) and args.projector == "default":
        rf = line["roofline"]
        if time.time() - t_main > LIVE_TRAFFIC_LATEST_START_S:
            live, how = None, "time: the run had already taken %.0f s (limit %.0f s for starting the counter passes)" % (
                time.time() - t_main, LIVE_TRAFFIC_LATEST_START_S)
        else:
            hv = os.environ.get("HIP_VISIBLE_DEVICES")
            dev_index = None
            if env.world > 1:   # the physical index of this rank's device
                dev_index = hv.split(",")[env.local] if hv else env.local
            live, how = live_traffic(head, rf["counter_kernel_match"], device_index=dev_index)
            if live is not None and env.world > 1:
                how += "; N = %d: the passes ran on rank 0's GPU at one rank's shard size" % env.world
        if live is not None:
            rf["traffic_static"] = {"traffic": rf.get("traffic"), "traffic_source": rf.get("traffic_source")}
            rf["traffic"], rf["traffic_source"] = live, how
        else:
            rf["traffic_live"] = "not measured: " + how
    detail_line = None
    if line is not None:
        # TWO lines: the long record (every workload's full dictionary) first, prefixed so that nothing mistakes
        # it for the contract line, and then ONE short line that starts with {"metric" (summary_line)
        detail_line = json.dumps(line)
        line = summary_line(line)
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
        env.long_barrier()
        env.dist.destroy_process_group()
    if out_line is not None:
        write_detail(detail_line)
        print(out_line, flush=True)


if __name__ == "__main__":
    main()
