"""CPU test of the N > 1 path: world_size 2 over gloo.

The row-sharded algorithm (y-sized data local, x-sized data
replicated, sum all-reduce of A_k^T y_k partials, of the Gram matrix and of the
row-sums) is run by two processes through the oracle's sharded entry with
torch.distributed all_reduce as the collective, and must reproduce the
single-process solve.  The HIP engine uses the same decomposition with RCCL in
place of gloo (pogs_amd/csrc/dense_solver.h: finish_cols / allreduce call sites).
Also covers the launcher-side plumbing bench.py uses: per-rank shard generation
and broadcasting an opaque 128-byte id from rank 0.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
from helpers import relerr, soa
from pogs_amd import graph as G
from pogs_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(m, n, dtype, sparse):
    if sparse:
        return synth.csr_lasso(m, n, 12, seed=31, dtype=dtype)
    return synth.dense_lasso(m, n, seed=31, dtype=dtype)


def _worker(rank, world, port, m, n, dtype_name, out_dir, sparse=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = np.dtype(dtype_name).type
    A, b, _ = _problem(m, n, dtype, sparse)
    rows = m // world
    lo, hi = rank * rows, (rank + 1) * rows if rank < world - 1 else m
    f, g = G.lasso_functions(b, 0.1, n)

    def allreduce(arr):
        t = torch.from_numpy(arr)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    # opaque id broadcast, as bench.py does for the RCCL unique id
    uid = torch.arange(128, dtype=torch.uint8) if rank == 0 else torch.zeros(128, dtype=torch.uint8)
    dist.broadcast(uid, 0)
    assert bytes(uid.tolist()) == bytes(range(128))

    r = ob.oracle_solve_shard(A[lo:hi], m, soa(f.slice(lo, hi)), soa(g), allreduce, dtype=dtype)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=r["x"], y=r["y"], l=r["l"], optval=r["optval"],
             iterations=r["iterations"], status=r["status"], lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 2e-4)])
def test_row_sharded_solve_matches_single_process(tmp_path, dtype, tol):
    m, n, world = 900, 120, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, m, n, np.dtype(dtype).name, str(tmp_path)), nprocs=world, join=True)
    A, b, _ = synth.dense_lasso(m, n, seed=31, dtype=dtype)
    f, g = G.lasso_functions(b, 0.1, n)
    want = ob.oracle_solve(A, soa(f), soa(g), dtype=dtype)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for p in parts:
        assert int(p["status"]) == want["status"] == 0
        assert abs(int(p["iterations"]) - want["iterations"]) <= (0 if dtype == np.float64 else 3)
        assert relerr(p["x"], want["x"]) < tol               # x is replicated
        assert float(p["optval"]) == pytest.approx(want["optval"], rel=max(tol, 1e-9))
    y = np.concatenate([p["y"] for p in parts])                  # y, l are sharded by rows
    l = np.concatenate([p["l"] for p in parts])
    assert relerr(y, want["y"]) < tol and relerr(l, want["l"]) < 10 * tol
    # replicas took identical decisions
    assert np.array_equal(parts[0]["x"], parts[1]["x"])


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-8), (np.float32, 5e-4)])
def test_row_sharded_sparse_cgls_matches_single_process(tmp_path, dtype, tol):
    """CSR row blocks + CGLS; A^T r partials and |q|^2 are all-reduced."""
    m, n, world = 1200, 300, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, m, n, np.dtype(dtype).name, str(tmp_path), True), nprocs=world, join=True)
    A, b, _ = _problem(m, n, dtype, True)
    f, g = G.lasso_functions(b, 0.1, n)
    want = ob.oracle_solve(A, soa(f), soa(g), dtype=dtype)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for p in parts:
        assert int(p["status"]) == want["status"] == 0
        assert abs(int(p["iterations"]) - want["iterations"]) <= (1 if dtype == np.float64 else 5)
        assert relerr(p["x"], want["x"]) < tol
        assert float(p["optval"]) == pytest.approx(want["optval"], rel=max(tol, 1e-8))
    y = np.concatenate([p["y"] for p in parts])
    assert relerr(y, want["y"]) < tol
    assert np.array_equal(parts[0]["x"], parts[1]["x"])
