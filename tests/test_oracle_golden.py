"""CPU tests: the oracle (oracle/pogs_oracle.cpp) against the golden fixtures that the
compiled reference produced (tests/golden/make_golden.py), the known answers the
reference's own tests pin, and the libstdc++ random-number known answers.

This is what "pins" the oracle; the GPU parity tests then compare the HIP engine
with the oracle.
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
from helpers import PROBLEMS, relerr, soa
from pogs_amd import graph as G
from pogs_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = np.load(os.path.join(GOLD, "reference_outputs.npz"))


def gold(prefix):
    return {k: REF[prefix + k] for k in ("x", "y", "l", "optval", "iterations", "status")}


def check(got, want, xtol, it_slack=0, opt_rel=1e-9):
    assert got["status"] == int(want["status"])
    assert abs(got["iterations"] - int(want["iterations"])) <= it_slack, (got["iterations"], int(want["iterations"]))
    assert relerr(got["x"], want["x"]) < xtol
    assert relerr(got["y"], want["y"]) < xtol
    assert relerr(got["l"], want["l"]) < 10 * xtol
    assert got["optval"] == pytest.approx(float(want["optval"]), rel=opt_rel)


def test_c1_readme_lasso_golden():
    """C1 exactly: status 0, iterations 100, optval 91.767119316812...."""
    A, b, lam = synth.readme_lasso()
    f, g = G.lasso_functions(b, lam, 300)
    want = gold("c1_f64_")
    assert int(want["iterations"]) == 100 and int(want["status"]) == 0
    assert float(want["optval"]) == pytest.approx(91.76711931681265, rel=1e-12)
    check(ob.oracle_solve(A, soa(f), soa(g), dtype=np.float64), want, 1e-10, 0, 1e-12)
    check(ob.oracle_solve(A, soa(f), soa(g), dtype=np.float32), gold("c1_f32_"), 2e-5, 2, 1e-5)


def test_final_iter_and_status_codes_golden():
    """max_iter = 5 -> status 3 (POGS_MAX_ITER), final_iter 4 (SURVEY.md findings 1-2)."""
    A, b, lam = synth.readme_lasso()
    f, g = G.lasso_functions(b, lam, 300)
    want = gold("c1_maxiter5_")
    assert int(want["status"]) == 3 and int(want["iterations"]) == 4
    got = ob.oracle_solve(A, soa(f), soa(g), max_iter=5)
    check(got, want, 1e-10, 0, 1e-10)


@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
@pytest.mark.parametrize("problem", list(PROBLEMS))
def test_problem_families_golden(problem, tag, dtype):
    rng = np.random.default_rng(7)
    m, n = 200, 100
    A = rng.standard_normal((m, n))
    b = A @ (rng.standard_normal(n) * (rng.random(n) < 0.2)) + 0.1 * rng.standard_normal(m)
    f, g = PROBLEMS[problem](b, n)
    want = gold("fam_%s_%s_" % (problem, tag))
    got = ob.oracle_solve(A, soa(f), soa(g), dtype=dtype)
    if int(want["status"]) != 0:
        assert got["status"] == int(want["status"]) and got["iterations"] == int(want["iterations"])
        return
    if dtype == np.float64:
        check(got, want, 1e-8, 1, 1e-9)
    else:
        check(got, want, 1e-4, max(3, int(0.1 * int(want["iterations"]))), 1e-4)


def test_dense_fp32_lasso_golden():
    A, b, _ = synth.dense_lasso(2000, 300, seed=11, dtype=np.float32)
    f, g = G.lasso_functions(b, 0.1, 300)
    check(ob.oracle_solve(A, soa(f), soa(g), dtype=np.float32), gold("lasso2000_f32_"), 1e-4, 5, 1e-4)


@pytest.mark.parametrize("lam,prefix", [(0.01, "logit4000_f32_"), (0.0, "logit4000_l0_f32_")])
def test_dense_fp32_logistic_golden(lam, prefix):
    A, y, _ = synth.dense_logistic(4000, 200, seed=5, dtype=np.float32)
    f, g = G.logistic_functions(y, lam, 200)
    want = gold(prefix)
    got = ob.oracle_solve(A, soa(f), soa(g), dtype=np.float32)
    check(got, want, 2e-5, 0, 1e-5)   # same 86 iterations; measured 4.9e-6 (x), 3.8e-6 (y), 3.7e-7 (optval)


def test_sparse_csr_golden():
    A, b, _ = synth.csr_lasso(3000, 800, 20, seed=4, dtype=np.float64)
    f, g = G.lasso_functions(b, 0.1, 800)
    check(ob.oracle_solve(A, soa(f), soa(g), dtype=np.float64), gold("csr3000_f64_"), 1e-7, 1, 1e-8)
    A, b, _ = synth.csr_lasso(20000, 5000, 50, seed=3, dtype=np.float32)
    f, g = G.lasso_functions(b, 0.1, 5000)
    want = gold("csr20000_f32_")
    # same 128 iterations as the compiled reference; measured 1.5e-7 (x), 9.7e-8 (y), 1.5e-6 (l)
    check(ob.oracle_solve(A, soa(f), soa(g), dtype=np.float32), want, 2e-6, 0, 1e-6)


def test_wide_dense_golden():
    """m <= n: the A A^T branch of the direct projector (projector_direct_dense.cpp:128-135)."""
    A, b, _ = synth.dense_lasso(120, 300, seed=6)
    f, g = G.lasso_functions(b, 0.1, 300)
    check(ob.oracle_solve(A, soa(f), soa(g), dtype=np.float64), gold("wide120_f64_"), 1e-8, 1, 1e-9)


def test_prox_table_golden():
    """ProxEval / FuncEval of the reference header on a 16 x 3 x 3 x 16 grid, both precisions."""
    tab = np.load(os.path.join(GOLD, "prox_table.npz"))["table"]
    for is_float, dtype in ((0.0, np.float64), (1.0, np.float32)):
        rows = tab[tab[:, 0] == is_float]
        for rho in np.unique(rows[:, 7]):
            r = rows[rows[:, 7] == rho]
            objs = {"h": r[:, 1].astype(np.int32), "a": r[:, 2], "b": r[:, 3], "c": r[:, 4], "d": r[:, 5], "e": r[:, 6]}
            got = ob.oracle_prox(objs, rho, r[:, 8], dtype=dtype)
            want = r[:, 9]
            ok = np.isfinite(want)
            assert np.array_equal(np.isfinite(got), ok)
            np.testing.assert_allclose(got[ok].astype(np.float64), want[ok], rtol=0, atol=0)  # same code path: bitwise
        # FuncEval, element by element (sum of one)
        for row in rows[::7]:
            objs = {k: np.array([v]) for k, v in zip("abcde", row[2:7])}
            objs["h"] = np.array([int(row[1])], np.int32)
            got = ob.oracle_func(objs, np.array([row[8]]), dtype=dtype)
            if np.isfinite(row[10]):
                assert got == pytest.approx(row[10], rel=1e-15 if dtype == np.float64 else 1e-7, abs=0)
            else:
                assert not np.isfinite(got) or np.isnan(row[10])


def test_proj_subgrad_table_golden():
    """ProjSubgradEval of the reference header (prox_lib.h:468-493) on a 16 x 4 x 11 x 7 grid that
    sits on and around every kink, incl. the a = 0 / c = 0 shortcuts; both precisions."""
    tab = np.load(os.path.join(GOLD, "projsub_table.npz"))["table"]
    assert tab.shape == (2 * 16 * 4 * 11 * 7, 10)
    for is_float, dtype in ((0.0, np.float64), (1.0, np.float32)):
        r = tab[tab[:, 0] == is_float]
        objs = {"h": r[:, 1].astype(np.int32), "a": r[:, 2], "b": r[:, 3], "c": r[:, 4], "d": r[:, 5], "e": r[:, 6]}
        got = ob.oracle_proj_subgrad(objs, r[:, 7], r[:, 8], dtype=dtype).astype(np.float64)
        want = r[:, 9]
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        assert np.array_equal(got[ok], want[ok])   # same arithmetic, same compiler: bitwise (infinities included)


def test_prox_known_answers_of_reference_tests():
    """tests/test_proximal.cpp:12-220 (closed-form raw prox values)."""
    F = G.Function
    cases = [(F.kZero, 5.0, 1.0, 5.0), (F.kIdentity, 5.0, 2.0, 4.5), (F.kAbs, 2.0, 2.0, 1.5), (F.kAbs, 0.3, 2.0, 0.0),
             (F.kAbs, -2.0, 2.0, -1.5), (F.kAbs, 0.5, 2.0, 0.0), (F.kSquare, 6.0, 3.0, 4.5),
             (F.kSquare, -4.0, 3.0, -3.0), (F.kIndEq0, 5.0, 1.0, 0.0), (F.kIndGe0, -2.0, 1.0, 0.0),
             (F.kIndLe0, 2.0, 1.0, 0.0), (F.kIndBox01, 1.5, 1.0, 1.0), (F.kIndBox01, -0.5, 1.0, 0.0),
             (F.kMaxPos0, 3.0, 2.0, 2.5), (F.kMaxPos0, 0.3, 2.0, 0.0), (F.kMaxPos0, -1.0, 2.0, -1.0),
             (F.kMaxNeg0, -3.0, 2.0, -2.5), (F.kMaxNeg0, -0.3, 2.0, 0.0), (F.kMaxNeg0, 1.0, 2.0, 1.0),
             (F.kHuber, 0.5, 2.0, 0.5 * 2.0 / 3.0), (F.kHuber, 5.0, 2.0, 4.5), (F.kHuber, -5.0, 2.0, -4.5)]
    for h, v, rho, want in cases:
        assert ob.oracle_prox_raw(h, v, rho) == pytest.approx(want, abs=1e-12)
    for v in (2.0, 0.0, -1.0):  # optimality of ProxExp (test_proximal.cpp:222-247)
        r = ob.oracle_prox_raw(F.kExp, v, 1.0)
        assert r + np.exp(r) == pytest.approx(v, abs=1e-6)
    assert ob.oracle_prox_raw(F.kAbs, 3.0, 2.0, np.float32) == pytest.approx(ob.oracle_prox_raw(F.kAbs, 3.0, 2.0),
                                                                              rel=1e-5)


def test_norm_est_start_vector_known_answers():
    """libstdc++ default_random_engine + uniform_real_distribution."""
    f = ob.oracle_rand(5, np.float32)
    d = ob.oracle_rand(5, np.float64)
    np.testing.assert_allclose(f, [7.82590359e-06, 0.131537795, 0.75560534, 0.458650142, 0.532767236], rtol=1e-7)
    np.testing.assert_allclose(d, [0.13153778773876065, 0.4586501320232198, 0.21895918621247895,
                                   0.67886471674068549, 0.93469289622673879], rtol=1e-15)
    # the engine's host-side restatement must be the same generator
    import pogs_amd

    assert np.array_equal(pogs_amd.rand_uniform(1000, np.float32), ob.oracle_rand(1000, np.float32))
    assert np.array_equal(pogs_amd.rand_uniform(1000, np.float64), ob.oracle_rand(1000, np.float64))
