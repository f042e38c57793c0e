"""Shared helpers for the parity tests."""
import numpy as np

import pogs_amd
from pogs_amd import graph as G


def soa(fv):
    """FunctionVector -> dict of arrays as oracle_binding expects."""
    return {k: getattr(fv, k) for k in "habcde"}


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


PROBLEMS = {
    "lasso": lambda b, n: G.lasso_functions(b, 0.1, n),
    "ridge": lambda b, n: G.ridge_functions(b, 0.5, n),
    "elastic_net": lambda b, n: G.elastic_net_functions(b, 0.1, 0.2, n),
    "logistic": lambda b, n: G.logistic_functions(np.sign(b) + (b == 0), 0.01, n),
    "logistic0": lambda b, n: G.logistic_functions(np.sign(b) + (b == 0), 0.0, n),
    "huber": lambda b, n: G.huber_functions(b, 1.0, 0.05, n),
    "svm": lambda b, n: G.svm_functions(np.sign(b) + (b == 0), 1.0, n),
    "nonneg_ls": lambda b, n: G.nonneg_ls_functions(b, n),
}


def objective(A, f, g, x):
    """sum f(Ax) + sum g(x) evaluated in float64 with numpy (independent of every engine)."""
    y = np.asarray(A @ x, np.float64).ravel()
    return _fsum(f, y) + _fsum(g, np.asarray(x, np.float64))


def _fsum(fv, v):
    h = fv.h
    a, b, c, d, e = fv.a, fv.b, fv.c, fv.d, fv.e
    t = a * v - b
    out = np.zeros_like(v)
    F = G.Function
    for code in np.unique(h):
        msk = h == code
        z = t[msk]
        if code == F.kAbs:
            r = np.abs(z)
        elif code == F.kSquare:
            r = 0.5 * z * z
        elif code == F.kLogistic:
            r = np.logaddexp(0, z)
        elif code == F.kHuber:
            r = np.where(np.abs(z) < 1, 0.5 * z * z, np.abs(z) - 0.5)
        elif code == F.kMaxPos0:
            r = np.maximum(z, 0)
        elif code == F.kMaxNeg0:
            r = np.maximum(-z, 0)
        elif code == F.kIdentity:
            r = z
        elif code in (F.kZero, F.kIndGe0, F.kIndLe0, F.kIndEq0, F.kIndBox01):
            r = np.zeros_like(z)
        else:
            raise NotImplementedError(code)
        out[msk] = r
    return float(np.sum(c * out + d * v + 0.5 * e * v * v))


def run_row_sharded(pogs, A, f, g, world, dtype, solver_kw=None, count_collectives=False, transport="1", bounds=None,
                    **solve_kw):
    """Solves with `world` ranks inside this process: one thread + one Solver per rank, rows split
    evenly, joined by the engine's in-process test communicator ("POGSLOCAL:" unique id, see
    pogs_amd/csrc/dist.h).  Verifies the engine's own row-sharded decomposition on ONE GPU.
    Returns the per-rank result dicts (x replicated, y / l row slices)."""
    import os
    import threading

    import numpy as np

    # the in-process communicator is refused without it; "1": stream-ordered (device slots + events, no
    # stream is ever waited for), "host": staged through the host (pogs_amd/csrc/dist.h)
    os.environ["POGS_AMD_TEST_TRANSPORT"] = transport
    m = A.shape[0]
    uid = (b"POGSLOCAL:" + os.urandom(8).hex().encode()).ljust(128, b"\0")
    # rows split evenly unless the caller names the shard boundaries (unequal shards)
    bounds = np.linspace(0, m, world + 1).astype(int) if bounds is None else np.asarray(bounds, int)
    assert len(bounds) == world + 1 and bounds[0] == 0 and bounds[-1] == m
    results, errors = [None] * world, []

    def work(r):
        try:
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            with pogs.Solver(A[lo:hi], dtype=dtype, dist=(r, world, m, uid), **(solver_kw or {})) as s:
                results[r] = s.solve(f.slice(lo, hi), g, **solve_kw)
                if count_collectives:
                    # a second solve on the same handle (factorisation cached): its all-reduce calls
                    # are the iteration loop's alone
                    st0 = s.stats()
                    again = s.solve(f.slice(lo, hi), g, **solve_kw)
                    st1 = s.stats()
                    results[r]["loop_collectives"] = dict(
                        calls=st1["collectives"] - st0["collectives"], iterations=int(again["iterations"]) + 1,
                        misses=int(st1["spec_misses"] - st0["spec_misses"]), hits=int(st1["spec_hits"] - st0["spec_hits"]))
        except Exception as e:  # pragma: no cover - surfaced below
            errors.append((r, e))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    assert all(r is not None for r in results)
    return results, bounds


def run_sharded_oracle(A, f, g, world, dtype, bounds=None, **solve_kw):
    """The ORACLE's row-sharded entry (oracle/pogs_oracle.cpp: OraclePogsShard* / OraclePogsSparseShard*)
    with `world` ranks as threads of this process and an in-test sum in rank order as the collective.
    Returns (per-rank result dicts, bounds) with the same row split as run_row_sharded."""
    import threading

    import oracle_binding as ob

    m = A.shape[0]
    bounds = np.linspace(0, m, world + 1).astype(int) if bounds is None else np.asarray(bounds, int)
    bar = threading.Barrier(world)
    slots = [None] * world
    results, errors = [None] * world, []

    def make_allreduce(r):
        def allreduce(arr):
            slots[r] = arr.copy()
            bar.wait(600)
            total = slots[0].copy()
            for q in range(1, world):
                total += slots[q]
            bar.wait(600)   # nobody overwrites a slot that is still being read
            arr[:] = total
        return allreduce

    def work(r):
        try:
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            results[r] = ob.oracle_solve_shard(A[lo:hi], m, soa(f.slice(lo, hi)), soa(g), make_allreduce(r), dtype=dtype,
                                               **solve_kw)
        except Exception as e:  # pragma: no cover - surfaced below
            errors.append((r, e))
            bar.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(900)
    assert not errors, errors
    assert all(r is not None for r in results)
    return results, bounds
