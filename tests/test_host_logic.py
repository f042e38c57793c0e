"""CPU tests of the host-side mirror of python/pogs/graph.py: the function encodings
of the seven solve_* problems equal, array for array, what the reference's own
Python layer hands to PogsD (captured in tests/golden/python_layer.npz), the API
surface is the same, and input validation behaves like the reference."""
import inspect
import os

import numpy as np
import pytest

import pogs_amd
from pogs_amd import graph as G

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

PL = np.load(os.path.join(os.path.dirname(__file__), "golden", "python_layer.npz"))
A, b = PL["A"], PL["b"]
lab = np.sign(b)
n = A.shape[1]

CASES = {
    "lasso": lambda: G.lasso_functions(b, 0.1, n),
    "ridge": lambda: G.ridge_functions(b, 0.5, n),
    "elastic_net": lambda: G.elastic_net_functions(b, 0.1, 0.2, n),
    "logistic": lambda: G.logistic_functions(lab, 0.01, n),
    "logistic0": lambda: G.logistic_functions(lab, 0.0, n),
    "huber": lambda: G.huber_functions(b, 1.5, 0.05, n),
    "svm": lambda: G.svm_functions(lab, 1.0, n),
    "nonneg_ls": lambda: G.nonneg_ls_functions(b, n),
}


@pytest.mark.parametrize("name", list(CASES))
def test_encodings_equal_reference_python_layer(name):
    f, g = CASES[name]()
    fa, ga = f.arrays(np.float64), g.arrays(np.float64)
    for k in "abcde":
        np.testing.assert_array_equal(fa[k], PL["%s_f_%s" % (name, k)])
        np.testing.assert_array_equal(ga[k], PL["%s_g_%s" % (name, k)])
    np.testing.assert_array_equal(fa["h"], PL["%s_f_h" % name])
    np.testing.assert_array_equal(ga["h"], PL["%s_g_h" % name])
    # defaults the reference passes down: rho, abs_tol, rel_tol, max_iter, verbose, adaptive_rho, gap_stop
    np.testing.assert_allclose(PL["%s_scalars" % name], [1.0, 1e-4, 1e-4, 2500, 0, 1, 1])


def test_elastic_net_quirk_is_reproduced():
    """graph.py:522 passes e = lambda2 / 2 while the engine's term is e x^2 / 2."""
    _, g = G.elastic_net_functions(b, 0.1, 0.2, n)
    assert np.all(g.e == 0.1) and np.all(g.c == 0.1) and np.all(g.h == int(G.Function.kAbs))


def test_solve_signatures_match_reference():
    expect = {
        "solve_lasso": ["A", "b", "lambd", "abs_tol", "rel_tol", "max_iter", "verbose", "rho"],
        "solve_ridge": ["A", "b", "lambd", "abs_tol", "rel_tol", "max_iter", "verbose", "rho"],
        "solve_elastic_net": ["A", "b", "lambda1", "lambda2", "abs_tol", "rel_tol", "max_iter", "verbose", "rho"],
        "solve_logistic": ["A", "b", "lambd", "abs_tol", "rel_tol", "max_iter", "verbose", "rho"],
        "solve_huber": ["A", "b", "delta", "lambd", "abs_tol", "rel_tol", "max_iter", "verbose", "rho"],
        "solve_svm": ["A", "b", "lambd", "abs_tol", "rel_tol", "max_iter", "verbose", "rho"],
        "solve_nonneg_ls": ["A", "b", "abs_tol", "rel_tol", "max_iter", "verbose", "rho"],
    }
    for name, params in expect.items():
        sig = inspect.signature(getattr(pogs_amd, name))
        got = list(sig.parameters)
        assert got[:len(params)] == params and got[len(params):] == ["dtype"]
        assert sig.parameters["abs_tol"].default == 1e-4 and sig.parameters["rel_tol"].default == 1e-4
        assert sig.parameters["max_iter"].default == 2500 and sig.parameters["rho"].default == 1.0
    assert inspect.signature(pogs_amd.solve_logistic).parameters["lambd"].default == 0.0
    assert inspect.signature(pogs_amd.solve_svm).parameters["lambd"].default == 1.0
    assert inspect.signature(pogs_amd.solve_huber).parameters["delta"].default == 1.0


def test_function_vector_and_objects_agree():
    objs = [G.FunctionObj(G.Function.kSquare, 1.0, float(bi), 1.0) for bi in b]
    fv = G.FunctionVector.from_objs(objs)
    f, _ = G.lasso_functions(b, 0.1, n)
    for k in "habcde":
        np.testing.assert_array_equal(getattr(fv, k), getattr(f, k))
    sl = f.slice(3, 9)
    assert len(sl) == 6 and np.array_equal(sl.b, b[3:9])


def test_length_validation_like_reference():
    f, g = G.lasso_functions(b, 0.1, n)
    with pytest.raises(AssertionError):  # graph.py:292-293
        G._solve_graph_form(A, f.slice(0, 5), g)
    with pytest.raises(ValueError):
        G._solve_graph_form(A, f, g, dtype=np.int32)


def test_bench_self_launches_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus 8` without a launcher re-executes itself under torch.distributed.run
    (one rank per GPU, 127.0.0.1 rendezvous); under a launcher (WORLD_SIZE set) it does not."""
    import importlib.util
    import os
    import sys
    import types

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []
    monkeypatch.setattr(bench.os, "execve", lambda exe, argv, env: calls.append((exe, argv, env)))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.maybe_spawn(types.SimpleNamespace(gpus=8))
    assert len(calls) == 1
    exe, argv, env = calls[0]
    assert exe == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in argv and argv[argv.index("--nproc-per-node") + 1] == "8"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # single GPU or already under a launcher: nothing happens
    bench.maybe_spawn(types.SimpleNamespace(gpus=1))
    monkeypatch.setenv("WORLD_SIZE", "8")
    bench.maybe_spawn(types.SimpleNamespace(gpus=8))
    assert len(calls) == 1


# ---- CVXPY front end (reference: python/pogs/cvxpy.py) ------------------------------------------

def test_cvxpy_detection_matches_the_reference_detector():
    """pogs_amd.cvxpy.detect_graph_form on the stand-in expression trees against what the
    REFERENCE's _detect_graph_form returned for the same trees (tests/golden/cvxpy_detection.json,
    generated by tests/golden/make_cvxpy_golden.py with the stand-ins installed as `cvxpy`):
    same pattern, lambda, optimum scale, A and b -- and None for the same problems."""
    import json

    import cvxpy_standins as S
    from pogs_amd import cvxpy as front

    want = json.load(open(os.path.join(ROOT, "tests", "golden", "cvxpy_detection.json")))
    problems = S.cases()
    assert set(problems) == set(want) and sum(v is not None for v in want.values()) >= 7
    for name, problem in problems.items():
        got = front.detect_graph_form(problem)
        if want[name] is None:
            assert got is None, name
            continue
        kind, params = got
        assert kind == want[name]["type"], name
        assert params.get("lambd") == want[name]["lambd"], name          # same arithmetic: bit-equal
        assert params.get("optval_scale") == want[name]["optval_scale"], name
        assert np.array_equal(np.asarray(params["A"]), np.asarray(want[name]["A"])), name
        assert np.array_equal(np.asarray(params["b"]), np.asarray(want[name]["b"])), name


def test_pogs_solve_fills_in_the_problem_or_falls_through(monkeypatch):
    """pogs_solve (python/pogs/cvxpy.py:33-92): on a detected pattern the variable value, status and
    the SCALED optimum are set and returned; on no pattern or a failed solve the problem's own
    solve() is called with the options.  (Solver calls are replaced here; the GPU suite runs them.)"""
    import cvxpy_standins as S
    from pogs_amd import cvxpy as front

    calls = []

    def fake(status):
        def solve(A, b, *lam, **kw):
            calls.append((A.shape, tuple(lam), kw))
            return dict(x=np.arange(A.shape[1], dtype=float), optval=3.0, status=status, iterations=7)
        return solve

    monkeypatch.setattr(front, "solve_lasso", fake(0))
    monkeypatch.setattr(front, "solve_ridge", fake(0))
    monkeypatch.setattr(front, "solve_nonneg_ls", fake(3))       # max-iter: "failed"
    cases = S.cases()
    p = cases["ridge"]                                           # 2 |Ax-b|^2 + 0.6 |x|^2 -> lambda 0.3, scale 4
    assert front.pogs_solve(p, max_iter=11, rho=2.0) == 12.0
    assert p._status == "optimal" and p._value == 12.0 and np.array_equal(p.variables()[0].value, np.arange(5.0))
    assert calls[-1][1] == (0.3,) and calls[-1][2]["max_iter"] == 11 and calls[-1][2]["rho"] == 2.0
    assert calls[-1][2]["abs_tol"] == 1e-4 and calls[-1][2]["rel_tol"] == 1e-4 and calls[-1][2]["verbose"] == 0
    assert not p.fallback_calls
    p = cases["nnls"]                                            # solver reports status 3 -> default solver
    front.pogs_solve(p, verbose=True)
    assert p.fallback_calls == [dict(verbose=True)] and p.variables()[0].value is None
    p = cases["plain_least_squares"]                             # no pattern -> default solver, options passed on
    front.pogs_solve(p, eps=1e-9)
    assert p.fallback_calls == [dict(verbose=False, eps=1e-9)]
    import pogs_amd

    assert pogs_amd.pogs_solve is front.pogs_solve               # exported like the reference's (__init__.py:29)


def test_bench_bookkeeping_without_a_gpu(tmp_path, monkeypatch):
    """bench.py's host-side pieces that need no GPU: the kernel-source hash that dates the committed
    PMC traffic figures (fresh / stale / unknown), the fixture the c2 workload is generated from, and
    the synthetic generators' reproducibility (the fixtures depend on it bit for bit)."""
    import importlib.util
    import json

    from pogs_amd import synth

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sha = bench.csrc_sha16()
    assert len(sha) == 16 and sha == bench.csrc_sha16()
    # a traffic file collected on these sources, on other sources, and one without a hash
    prof = tmp_path / "profiles"
    prof.mkdir()
    entry = {"kern<float, 1>": {"launches": 4, "hbm_bytes_per_launch_corrected": 10.0},
             "kern<float, 2>": {"launches": 1, "hbm_bytes_per_launch_corrected": 20.0},
             "other": {"launches": 9, "hbm_bytes_per_launch_corrected": 99.0}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_sha16", lambda: sha)
    for meta, want in (({"csrc_sha16": sha}, True), ({"csrc_sha16": "0" * 16}, False), (None, None)):
        d = dict(entry)
        if meta is not None:
            d["_meta"] = meta
        (prof / "pmc_traffic_c9.json").write_text(json.dumps(d))
        val, rel, fresh = bench.pmc_traffic("c9", "kern<float")
        assert val == pytest.approx((4 * 10.0 + 20.0) / 5) and rel.endswith("pmc_traffic_c9.json") and fresh is want
    assert bench.pmc_traffic("nope", "kern") == (None, None, None)
    # the c2 fixture names the generator's arguments; a small instance of that generator is reproducible
    fx = np.load(os.path.join(ROOT, "tests", "golden", "c2_reference.npz"))
    assert tuple(int(v) for v in fx["shape"]) == (bench.CONFIGS["c2"]["m"], bench.CONFIGS["c2"]["n"]) and int(fx["seed"]) == 2024
    A1, b1, _ = synth.dense_lasso_rows(900, 70, seed=int(fx["seed"]), chunk=400)
    A2, b2, _ = synth.dense_lasso_rows(900, 70, seed=int(fx["seed"]), chunk=400)
    assert np.array_equal(A1, A2) and np.array_equal(b1, b2) and A1.dtype == np.float32
    L1 = synth.dense_logistic_rows(500, 40, seed=33, chunk=200)
    L2 = synth.dense_logistic_rows(500, 40, seed=33, chunk=200)
    assert np.array_equal(L1[0], L2[0]) and np.array_equal(L1[1], L2[1]) and set(np.unique(L1[1])) <= {-1.0, 1.0}


def test_bench_extrapolates_the_cpu_leg_of_a_multi_gpu_line():
    """bench.py at N > 1: rank 0 times the reference on its own shard and the line states the whole problem's
    rate as an extrapolation t_iter ~ m; at N = 1 nothing is touched."""
    import sys

    sys.path.insert(0, ROOT)
    import bench

    base = {"value": 16.0, "unit": "it/s", "time_to_converge_s": 19.0, "init_s": 12.0, "loop_s": 7.0, "sample": "whole workload"}
    one = bench.extrapolate_cpu(dict(base), 1, 100000, 10000)
    assert one == base
    r = bench.extrapolate_cpu(dict(base), 8, 100000, 10000)
    assert r["value"] == 2.0 and r["value_on_one_shard"] == 16.0 and r["value_in_metric_unit"] == 16.0
    assert "800000 x 10000" in r["unit"] and "t_iter ~ m" in r["extrapolated"] and "gsl_vector.h:135-138" in r["extrapolated"]
    assert "time_to_converge_s" not in r and r["time_to_converge_s_on_one_shard"] == 19.0
    assert r["sample"].startswith("rank 0's shard")
    assert bench.extrapolate_cpu({"value": None}, 8, 1, 1) == {"value": None}


def test_bench_windows_cover_whole_solves():
    """bench.py times exactly K steps per window; the windows together cover a whole number of solves, because the
    iterations of a solve do not cost the same (C4: 1.5 ms early, 0.53 ms in the middle, 1.06 ms at the end) and the
    metric is a solve's iterations over its loop time."""
    import sys

    sys.path.insert(0, ROOT)
    import bench

    for K, L, per_step in ((20, 359, 0.87e-3), (20, 106, 0.68e-3), (20, 189, 0.72e-3), (200, 359, 0.87e-3), (200, 106, 0.68e-3)):
        w, err = bench.pick_windows(K, L, per_step)
        solves = w * K / float(L)
        assert 1 <= w <= 40 and abs(solves - round(solves)) <= 0.06 and err <= 0.06, (K, L, w, solves)
        assert w * K * per_step >= 0.25
    assert bench.pick_windows(20, 359, 0.87e-3)[0] == 18                 # one C4 solve
    assert bench.pick_windows(1000, 106, 0.68e-3)[0] == 1               # K spans many solves already
    w, _ = bench.pick_windows(20, 2500, 1e-3)                          # a solve longer than 40 windows: as many as allowed
    assert w == 40


def test_bench_contract_line_is_short_and_starts_with_metric(capsys, tmp_path, monkeypatch):
    """BENCH_r05.parsed was null: a 24 KB line that began with `secondary`, of which the driver's 8 KB stdout
    tail held the end.  bench.summary_line turns a full record (here: round 5's committed one,
    profiles/r05_bench_driver_cmd.json) into the contract line -- first key `metric`, under 6 KB, `roofline` and
    `cpu_baseline` with the contract's fields, a short block per secondary workload -- and write_detail puts the
    long record on its own prefixed line BEFORE it."""
    import json

    import bench

    full = json.load(open(os.path.join(os.path.dirname(bench.__file__), "profiles", "r05_bench_driver_cmd.json")))
    full.pop("headline", None)
    sm = bench.summary_line(full)
    text = json.dumps(sm)
    assert text.startswith('{"metric"') and len(text) < bench.SUMMARY_MAX_CHARS <= 6000, len(text)
    assert "NaN" not in text and "Infinity" not in text
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in sm, key
    assert sm["config"]["name"] == "c2" and "configs[1]" in sm["config"]["workload"] and "model" not in sm["config"]
    rf = sm["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 and abs(rf["frac"] - full["roofline"]["frac"]) < 1e-5
    cb = sm["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] == "reference" and cb["cores"] == 16
    assert abs(sm["value"] - full["value"]) < 1e-5 * full["value"]
    assert abs(sm["value"] - 1e3 / sm["ms_per_step"]) < 1e-4 * sm["value"]
    assert sm["parity"]["iterations_engine"] == sm["parity"]["iterations_reference"] == 106
    for name in ("c3", "c4", "c2f64"):
        s, f = sm["secondary"][name], full["secondary"][name]
        assert abs(s["value"] - f["value"]) < 1e-5 * f["value"]
        assert abs(s["roofline"]["frac"] - f["roofline"]["frac"]) < 1e-5
        assert abs(s["parity_rel_x"] - f["parity_vs_reference"]["rel_x"]) <= 1e-5 * f["parity_vs_reference"]["rel_x"]
    # a record with oversized strings / lists still comes out under the limit
    fat = json.loads(json.dumps(full))
    fat["config"]["workload"] = "w" * 5000
    fat["cpu_baseline"]["sample"] = "s" * 5000
    fat["secondary"]["c3"]["roofline"]["kernel"] = "k" * 5000
    assert len(json.dumps(bench.summary_line(fat))) < bench.SUMMARY_MAX_CHARS
    # the long record goes out first, prefixed: it neither starts with "{" nor ends the output
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.write_detail(json.dumps(full))
    print(text)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 2 and out[0].startswith(bench.DETAIL_PREFIX) and out[1] == text
    assert json.loads(out[0][len(bench.DETAIL_PREFIX):]) == full
    assert json.load(open(tmp_path / "gpurun_out" / "bench_detail.json")) == full
