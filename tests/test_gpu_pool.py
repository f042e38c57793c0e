"""The library's device memory pool (csrc/common.h: DevicePool; include/pogs_amd.h: PogsAmdPoolStats /
PogsAmdPoolTrim).  The reference builds and destroys its solver inside every one-shot call
(src/interface_c/pogs_c.cpp:19-20), so the working set is allocated and freed per call; here the blocks
of a destroyed handle are handed to the next one, and that must change neither results nor -- beyond the
first handle -- the setup time."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle_binding as ob
from helpers import relerr, soa

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _pogs():
    import pogs_amd

    return pogs_amd


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def test_five_consecutive_handles_on_c2_set_up_at_the_speed_of_the_fastest():
    """configs[1]'s matrix, create / solve / destroy five times: every handle after the first takes its
    5.6 GB from the pool (no hipMalloc, no first touch, no hipFree), sets up within 1.3x of the fastest
    and returns the same bits."""
    torch = _torch()
    pogs = _pogs()
    from pogs_amd import _lib

    m, n = 100000, 10000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32)
    xt = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.1)
    b = A @ xt + 0.1 * torch.randn(m, generator=g, device=dev)
    torch.cuda.synchronize()
    f, gg = pogs.graph.lasso_functions(b.double().cpu().numpy(), 0.1, n)
    # one throw-away handle: the process's first one also pays for stream creation and code loading
    pogs.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True).close()
    def five_cycles():
        setups, totals, runs, pool = [], [], [], []
        for _ in range(5):
            p0 = _lib.pool_stats()
            t0 = time.time()
            s = pogs.Solver(A.data_ptr(), dtype=np.float32, shape=(m, n), device_ptr=True)
            setups.append(time.time() - t0)
            runs.append(s.solve(f, gg))
            s.close()
            totals.append(time.time() - t0)
            p1 = _lib.pool_stats()
            pool.append({k: p1[k] - p0[k] for k in ("mallocs", "reuses", "frees")})
        print("setup_s", setups, "create+solve+destroy_s", totals, "pool", pool)
        return setups, totals, runs, pool

    def steady(setups, totals):
        # 0.059 s of kernels + host; the stall this test was written against was 0.28 s, every time
        return max(setups) < 1.3 * min(setups) and max(totals) < 1.3 * min(totals) and max(setups) < 0.085

    setups, totals, runs, pool = five_cycles()
    if not steady(setups, totals):
        # one repetition: in a full-suite run of round 5 ONE of five set-ups took 0.120 s instead of 0.053 (all blocks from the
        # pool, no hipMalloc / hipFree; three isolated runs of this file: 0.054-0.055 throughout) -- a systematic stall fails twice
        # (the repetition is REPORTED -- a warning in the run's summary with the first attempt's figures and pool counters --
        # so that an intermittent stall stays visible instead of passing silently)
        import warnings

        warnings.warn("five-cycle timing was not steady on the first attempt (repeated once): setup_s %s, create+solve+destroy_s %s, "
                      "pool deltas %s" % ([round(v, 4) for v in setups], [round(v, 4) for v in totals], pool))
        setups, totals, runs, pool = five_cycles()
    assert max(setups) < 1.3 * min(setups), setups
    assert max(totals) < 1.3 * min(totals), totals
    assert max(setups) < 0.085, setups
    for d in pool[1:]:
        assert d["mallocs"] == 0 and d["frees"] == 0 and d["reuses"] > 0, pool
    for r in runs[1:]:
        assert r["status"] == 0 and r["iterations"] == runs[0]["iterations"]
        for k in "xyl":
            assert np.array_equal(r[k], runs[0][k]), k
    # the idle blocks can be given back on request
    freed = _lib.pool_trim()
    assert freed >= 4 * m * n
    assert _lib.pool_stats()["cached_bytes"] == 0


def test_one_shot_calls_reuse_their_blocks_and_match_the_oracle():
    """PogsS / PogsD build a handle per call (as the reference does): the second call of a shape runs
    entirely on recycled blocks and still matches the oracle -- dense and sparse, both types."""
    _torch()
    pogs = _pogs()
    from pogs_amd import _lib, synth

    for dtype in (np.float32, np.float64):
        A, b, _ = synth.dense_lasso(3000, 400, seed=5, dtype=dtype)
        f, g = pogs.graph.lasso_functions(b, 0.1, 400)
        want = ob.oracle_solve(A, soa(f), soa(g), dtype=dtype)
        first = pogs.solve_lasso(A, b, 0.1, dtype=dtype)
        p0 = _lib.pool_stats()
        second = pogs.solve_lasso(A, b, 0.1, dtype=dtype)
        p1 = _lib.pool_stats()
        assert p1["mallocs"] == p0["mallocs"] and p1["reuses"] > p0["reuses"]
        assert np.array_equal(first["x"], second["x"]) and first["iterations"] == second["iterations"]
        assert relerr(second["x"], want["x"]) < (1e-6 if dtype == np.float64 else 1e-4)
    import scipy.sparse as sp

    rng = np.random.default_rng(3)
    S = sp.random(6000, 1500, density=0.01, random_state=rng, format="csr", dtype=np.float64)
    b = S @ (rng.standard_normal(1500) * (rng.random(1500) < 0.1)) + 0.1 * rng.standard_normal(6000)
    f, g = pogs.graph.lasso_functions(b, 0.1, 1500)
    want = ob.oracle_solve(S, soa(f), soa(g), dtype=np.float64)
    first = pogs.solve_lasso(S, b, 0.1)
    p0 = _lib.pool_stats()
    second = pogs.solve_lasso(S, b, 0.1)
    p1 = _lib.pool_stats()
    assert p1["mallocs"] == p0["mallocs"] and p1["reuses"] > p0["reuses"]
    assert np.array_equal(first["x"], second["x"])
    assert relerr(second["x"], want["x"]) < 1e-6


_POISON_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
import pogs_amd
from pogs_amd import synth
import oracle_binding as ob
from helpers import PROBLEMS, relerr, soa
import scipy.sparse as sp

worst = 0.0
for (m, n) in ((700, 230), (230, 700), (4001, 333)):
    for dtype in (np.float64, np.float32):
        A, b, _ = synth.dense_lasso(m, n, seed=m, dtype=dtype)
        for fam in ("lasso", "logistic", "huber"):
            f, g = PROBLEMS[fam](b, n)
            want = ob.oracle_solve(A, soa(f), soa(g), dtype=dtype)
            for order in (pogs_amd.Ordering.ROW_MAJ, pogs_amd.Ordering.COL_MAJ):
                with pogs_amd.Solver(A, dtype=dtype, order=order) as s:
                    got = s.solve(f, g)
                assert got["status"] == want["status"], (m, n, fam, dtype)
                e = relerr(got["x"], want["x"])
                assert e < (1e-6 if dtype == np.float64 else 2e-4), (m, n, fam, dtype, order, e)
                worst = max(worst, e)
rng = np.random.default_rng(9)
for (m, n) in ((5000, 1200), (900, 2500)):
    S = sp.random(m, n, density=0.02, random_state=rng, format="csr", dtype=np.float64)
    b = S @ (rng.standard_normal(n) * (rng.random(n) < 0.1)) + 0.1 * rng.standard_normal(m)
    f, g = PROBLEMS["lasso"](b, n)
    want = ob.oracle_solve(S, soa(f), soa(g), dtype=np.float64)
    got = pogs_amd._solve_graph_form(S, f, g)
    assert got["status"] == want["status"]
    assert relerr(got["x"], want["x"]) < 1e-6
print("POISON_OK worst %%.2e" %% worst)
"""


def test_no_kernel_counts_on_fresh_memory_being_zero():
    """POGS_AMD_POOL_POISON=1 fills every block the pool hands out with 0xFF bytes (NaN): dense tall /
    wide / ragged shapes in both orderings, three function families, both types, and CSR must
    still follow the oracle -- i.e. every pad lane and scratch word the kernels read is written first."""
    _torch()
    env = dict(os.environ, POGS_AMD_POOL_POISON="1", POGS_AMD_TORCH_PRELOAD="1")
    r = subprocess.run([sys.executable, "-c", _POISON_SCRIPT % {"root": ROOT}], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "POISON_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_POOL_OFF_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch  # noqa: F401
import pogs_amd
from pogs_amd import _lib, synth
A, b, _ = synth.dense_lasso(3000, 400, seed=5, dtype=np.float32)
r1 = pogs_amd.solve_lasso(A, b, 0.1, dtype=np.float32)
r2 = pogs_amd.solve_lasso(A, b, 0.1, dtype=np.float32)
st = _lib.pool_stats()
assert np.array_equal(r1["x"], r2["x"]) and r1["status"] == 0
assert st["reuses"] == 0 and st["cached_bytes"] == 0 and st["frees"] == st["mallocs"] > 0, st
print("POOL_OFF_OK", st["mallocs"])
"""


def test_pool_can_be_turned_off():
    """POGS_AMD_POOL_MB=0: every block goes back to the HIP runtime when it is released (the behaviour
    before round 4), same results."""
    _torch()
    env = dict(os.environ, POGS_AMD_POOL_MB="0", POGS_AMD_TORCH_PRELOAD="1")
    r = subprocess.run([sys.executable, "-c", _POOL_OFF_SCRIPT % {"root": ROOT}], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "POOL_OFF_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
