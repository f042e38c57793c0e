"""The driver's bench command, exercised every round: the default invocation (c2 headline line with
`secondary` c3 / c4 / c2f64) and the same workloads through the RCCL code path with a one-rank communicator
(POGS_AMD_FORCE_DIST=1) -- the exact command the driver scales to N = 2, 4, 8."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    last = lines[-1]                                  # the contract line is the LAST line of stdout ...
    assert last.startswith('{"metric"') and len(last) < 6000, (len(last), last[:80])   # ... short, and starts with `metric`
    summary = json.loads(last)
    assert [ln for ln in lines if ln.startswith("{")] == [last]          # the only line a JSON scanner can pick up
    detail = [ln for ln in lines if ln.startswith("BENCH_DETAIL ")]
    assert len(detail) == 1 and lines.index(detail[0]) < len(lines) - 1
    d = json.loads(detail[0][len("BENCH_DETAIL "):])     # the long record: every workload's whole dictionary
    with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json")) as fh:
        assert json.load(fh) == d
    d["_summary"] = summary
    return d


def _close(a, b, rel=1e-5):
    return abs(a - b) <= rel * abs(b)


@pytest.fixture(scope="module")
def plain():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` as the driver runs it, GPU legs only."""
    return _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])


def test_default_invocation_carries_the_contract_fields_and_the_secondary_workloads(plain):
    d = plain
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "it/s"
    assert d["config"]["name"] == "c2" and "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    assert d["config"]["rccl_nranks"] == 0
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["scaling"] == "weak"
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0.5 < rf["frac"] < 1.0 and abs(rf["bytes_per_launch"] - 4.0e9) < 1e6
    # roofline.traffic comes from counters collected BY THIS RUN (two short rocprofv3 --pmc passes of the headline
    # workload after the timed part); the committed summary rides along for comparison
    # (on a box so cold that the run is already minutes old the passes are skipped and the line says so)
    if "traffic_live" in rf:
        assert rf["traffic_live"].startswith("not measured: time"), rf["traffic_live"]
    else:
        assert rf["traffic_source"].startswith("measured in this run"), rf["traffic_source"]
        assert "traffic_static" in rf
    assert 0.97 * 4.0e9 < rf["traffic"] < 1.10 * 4.0e9, rf["traffic"]
    # both peaks: the data-sheet 8 TB/s and this device's measured read ceiling
    assert rf["peak_datasheet"] == 8000.0 and 5000.0 < rf["peak_measured"] < 8000.0, rf["peak_measured"]
    assert abs(rf["frac_of_peak_measured"] - rf["achieved"] / rf["peak_measured"]) < 1e-12 and rf["frac_of_peak_measured"] < 1.02
    assert abs(rf["iteration_frac"] - rf["iteration"]["frac"]) < 1e-12
    assert abs(d["value"] - 1e3 * 1 / d["ms_per_step"]) <= 1e-6 * d["value"]
    # wall-clock-to-converge (the other half of BASELINE.json's metric) where the driver's record keeps it: in
    # `config`, and in the `headline` summary the line ENDS with
    cf = d["config"]
    assert cf["time_to_converge_s"] == d["time_to_converge_s"] and cf["init_s"] == d["init_s"]
    assert cf["handle_cycles_max_time_to_converge_s"] == d["handle_cycles"]["max_time_to_converge_s"]
    assert d["parity_vs_reference"]["rel_x"] == cf["parity_rel_x"]
    # the setup with its two shortcuts switched off (native fp32 Gram, all 50 Sinkhorn-Knopp passes): same solve,
    # a dearer setup, stated next to the default
    ex = d["exact_setup"]
    assert ex["status"] == 0 and abs(ex["iterations"] - d["solve_iterations"]) <= 2 and ex["rel_x_vs_default_setup"] < 5e-5, ex
    assert d["time_to_converge_exact_setup_s"] == ex["time_to_converge_s"] == cf["time_to_converge_exact_setup_s"]
    assert d["time_to_converge_s"] < ex["time_to_converge_s"] < 0.6, (d["time_to_converge_s"], ex["time_to_converge_s"])
    # the reference user's one-shot call with a HOST matrix, and the upload on its own
    os_ = d["one_shot_host_call"]
    assert os_["status"] == 0 and os_["iterations"] == d["solve_iterations"], os_
    assert 0.0 < os_["h2d_s"] < os_["one_shot_host_call_s"] < 3.0 and cf["h2d_s"] == os_["h2d_s"], os_
    assert os_["one_shot_host_call_s"] >= d["time_to_converge_s"]
    assert d["solve_status"] == 0 and 95 <= d["solve_iterations"] <= 117          # the fixture problem: 106
    sec = d["secondary"]
    for name, idx in (("c3", 2), ("c4", 3), ("c2f64", 1)):
        s = sec[name]
        assert s.get("value"), s
        assert "configs[%d]" % idx in s["config"]["workload"] and s["solve_status"] == 0
        assert 0.3 < s["roofline"]["frac"] < 1.0 and s["ms_per_step"] > 0
    assert sec["c2f64"]["dtype"] == "f64" and abs(sec["c2f64"]["roofline"]["bytes_per_launch"] - 8.0e9) < 1e6
    # every workload is the problem of a committed fixture of the compiled reference's own solution:
    # parity travels in the line (north star: x within 1e-4, the reference's iteration count +-10 %)
    for name, s in [("c2", d)] + [(k, sec[k]) for k in ("c3", "c4", "c2f64")]:
        par = s["parity_vs_reference"]
        assert par["rel_x"] < 1e-4, (name, par)
        assert abs(par["iterations_engine"] - par["iterations_reference"]) <= 0.1 * par["iterations_reference"], (name, par)
        assert "tests/golden/" in par["against"]
    assert sec["c2f64"]["parity_vs_reference"]["rel_x"] < 1e-9       # fp64 walks the reference's own trajectory
    # create / solve / destroy cycles: the device pool keeps every one at the speed of the fastest
    # (the round-3 driver run saw the second handle of a process set up in 0.342 s instead of 0.06 s)
    for name, s, init_cap, ttc_cap in (("c2", d, 0.075, 0.16), ("c3", sec["c3"], 0.05, 0.21), ("c4", sec["c4"], 0.09, 0.5)):
        hc = s["handle_cycles"]
        assert hc["n"] == 5 and hc["pool"]["hipMalloc_calls"] == 0 and hc["pool"]["hipFree_calls"] == 0, (name, hc["pool"])
        assert hc["max_init_s"] <= init_cap and hc["max_time_to_converge_s"] <= ttc_cap, (name, hc)
        assert hc["max_init_s"] <= 1.3 * min(hc["init_s"]), (name, hc["init_s"])
        assert s["init_s"] <= init_cap and s["time_to_converge_s"] <= ttc_cap, (name, s["init_s"], s["time_to_converge_s"])


def test_the_contract_line_is_short_and_carries_every_workloads_figures(plain):
    """BENCH_r05.parsed was null: the line had grown to 24 KB and began with `secondary`, the driver keeps an 8 KB
    tail.  The LAST line is now a summary (bench.summary_line): it starts with `metric`, stays under 6 KB, and
    still holds `roofline`, the wall-clock and parity scalars and a short block per secondary workload; the long
    record is the line before it (asserted in _bench) and gpurun_out/bench_detail.json."""
    d, sm = plain, plain["_summary"]
    assert list(sm)[0] == "metric" and len(json.dumps(sm)) < 6000
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "time_to_converge_s", "init_s", "parity", "secondary"):
        assert key in sm, key
    assert sm["metric"] == d["metric"] and sm["n_gpus"] == 1 and sm["steps"] == 20 and sm["warmup"] == 5
    assert sm["config"]["name"] == "c2" and "configs[1]" in sm["config"]["workload"] and "model" not in sm["config"]
    assert _close(sm["value"], d["value"]) and _close(sm["ms_per_step"], d["ms_per_step"])
    assert _close(sm["value"], 1e3 / sm["ms_per_step"], 1e-4)
    rf, rd = sm["roofline"], d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "peak_measured", "iteration_frac",
                "bytes_per_launch", "avg_launch_ms"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and _close(rf["frac"], rf["achieved"] / rf["peak"], 1e-4)
    assert _close(rf["frac"], rd["frac"]) and _close(rf["traffic"], rd["traffic"]) and _close(rf["iteration_frac"], rd["iteration_frac"])
    assert _close(sm["time_to_converge_s"], d["time_to_converge_s"]) and _close(sm["init_s"], d["init_s"])
    assert sm["solve_iterations"] == d["solve_iterations"] and sm["solve_status"] == 0
    assert _close(sm["parity"]["rel_x"], d["parity_vs_reference"]["rel_x"]) and sm["parity"]["iterations_engine"] == sm["parity"]["iterations_reference"] == 106
    assert _close(sm["handle_cycles_max_time_to_converge_s"], d["handle_cycles"]["max_time_to_converge_s"])
    assert _close(sm["one_shot_host_call_s"], d["one_shot_host_call"]["one_shot_host_call_s"])
    for name in ("c3", "c4", "c2f64"):
        s, full = sm["secondary"][name], d["secondary"][name]
        assert _close(s["value"], full["value"]) and _close(s["roofline"]["frac"], full["roofline"]["frac"])
        assert _close(s["roofline"]["iteration_frac"], full["roofline"]["iteration_frac"])
        assert _close(s["parity_rel_x"], full["parity_vs_reference"]["rel_x"]) and s["parity_rel_x"] < 1e-4
        assert s["solve_status"] == 0 and s["parity_iterations"].split()[0] == str(full["parity_vs_reference"]["iterations_engine"])
    assert "cpu_baseline" not in sm          # this fixture runs with --no-cpu-baseline (the forced-communicator test has it)


def test_forced_communicator_line_is_contract_complete():
    """The first multi-GPU line must not be the first time its extra legs run: the driver's
    command through the one-rank RCCL path carries `cpu_baseline`, a `roofline.traffic` measured in the run (the
    counter passes' children must not inherit the rendezvous / forced-communicator environment),
    `parity_vs_reference` and `config.rccl_nranks == 1`."""
    d = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-secondary", "--cpu-best-only"],
               {"POGS_AMD_FORCE_DIST": "1"}, timeout=900)
    assert d["config"]["rccl_nranks"] == 1 and d["n_gpus"] == 1 and d["config"]["name"] == "c2"
    cb = d["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb, cb
    rf = d["roofline"]
    if "traffic_live" in rf:
        assert rf["traffic_live"].startswith("not measured: time"), rf["traffic_live"]
    else:
        assert rf["traffic_source"].startswith("measured in this run"), rf["traffic_source"]
    assert 0.97 * 4.0e9 < rf["traffic"] < 1.10 * 4.0e9, rf["traffic"]
    par = d["parity_vs_reference"]
    assert par["rel_x"] < 1e-4 and abs(par["iterations_engine"] - par["iterations_reference"]) <= 3, par
    assert d["time_to_converge_s"] == d["config"]["time_to_converge_s"]
    sm = d["_summary"]
    assert sm["cpu_baseline"]["kind"] == "reference" and _close(sm["cpu_baseline"]["value"], cb["value"])
    assert sm["config"]["rccl_nranks"] == 1 and "secondary" not in sm and sm["parity"]["iterations_reference"] == par["iterations_reference"]


@pytest.mark.parametrize("cfg", ["c2", "c4"])
def test_one_rank_rccl_path_runs_the_bench_command_at_the_plain_speed(plain, cfg):
    """rccl_nranks is what ncclCommCount reports; one rank's collectives are no-ops, so the line must
    come out within a few per cent of the plain run -- for c4 too since round 4: row shards run the
    device-resident CGLS loop with ONE grouped all-reduce per CG step on the stream (t = A^T q and the
    |q|^2 records; A^T r by recurrence) and no host poll inside the projection: measured on one box,
    alternating, -1.5 / -2.2 / -1.9 % against the plain run (scripts/c4_one_rank.sh)."""
    d = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-live-traffic", "--config", cfg],
               {"POGS_AMD_FORCE_DIST": "1"})
    ref = plain if cfg == "c2" else plain["secondary"][cfg]
    assert d["config"]["rccl_nranks"] == 1 and d["n_gpus"] == 1
    assert d["solve_status"] == 0 and abs(d["solve_iterations"] - ref["solve_iterations"]) <= 3
    # the two runs are separate processes timing 20-step windows: c4's one-rank line was seen at -1.5 % .. -6.8 %
    # of the plain one over this round's runs of this test (-1.5 / -2.2 / -1.9 % with 100-step windows, alternating,
    # scripts/c4_one_rank.sh); round 3's host-polled loop was at -35 %
    slack = 0.05 if cfg == "c2" else 0.10
    assert d["value"] >= (1.0 - slack) * ref["value"], (d["value"], ref["value"])


def test_the_n_gt_1_parity_leg_on_two_in_process_ranks():
    """bench.py at N > 1: every rank draws its rows with torch_rows(seed 1000 + rank), rank 0 regenerates
    all of them and solves the whole problem unsharded (unsharded_parity).  No node here, so the two ranks
    are threads joined by the in-process, stream-ordered test communicator on one GPU -- the generator,
    the regeneration check and the comparison are the bench's own code."""
    import numpy as np
    import torch

    import bench
    import pogs_amd
    from helpers import run_row_sharded
    from pogs_amd import graph as G

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg, m, n, world = bench.CONFIGS["c2"], 12000, 1500, 2
    dev = torch.device("cuda:0")
    parts = [bench.torch_rows(cfg, m, n, r, dev) for r in range(world)]
    A = np.concatenate([p[0].cpu().numpy() for p in parts])
    b = np.concatenate([p[1] for p in parts])
    f, g = bench.functions(cfg, G, b, n)
    res, _ = run_row_sharded(pogs_amd, A, f, g, world, np.float32)
    par = bench.unsharded_parity(cfg, m, n, world, dev, 0, res[0], [float(p[1].sum()) for p in parts])
    assert par["rel_x"] < 1e-4 and abs(par["iterations_engine"] - par["iterations_reference"]) <= 3, par
    with pytest.raises(AssertionError):      # rows that are not the rank's own are noticed
        bench.unsharded_parity(cfg, m, n, world, dev, 0, res[0], [1.0, 2.0])


def test_two_process_rehearsal_of_the_launcher_path():
    """The first SCALE_r*.json will be produced by the driver on a node this build has never seen: everything
    around the collective must have run before.  Two PROCESSES under torch.distributed.run exactly as the driver
    starts them (`--nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`),
    both on cuda:0 -- RCCL refuses that, so the process group is gloo and the solver handles are joined by the test
    plug-in's shared-memory communicator (POGS_AMD_BENCH_REHEARSAL=1, tests/transport/test_transport.hip).  Runs at
    a reduced shape through: rank environment, the 128-byte id broadcast, per-rank shard generation, windows agreed by broadcast,
    max-over-ranks timing, the all-gather of the shards' checksums, rank 0's unsharded parity solve and CPU leg
    while rank 1 waits in long_barrier, and the assembly of the two output lines."""
    import socket
    import time

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, POGS_AMD_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "POGS_AMD_FORCE_DIST"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "20", "--warmup", "5", "--rows-per-gpu", "12000", "--cols", "1500"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    wall = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    last = lines[-1]
    assert last.startswith('{"metric"') and len(last) < 6000, last[:200]
    assert [ln for ln in lines if ln.startswith("{")] == [last]      # ONE contract line: rank 0's
    sm = json.loads(last)
    det = json.loads([ln for ln in lines if ln.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):])
    assert sm["n_gpus"] == 2 and sm["steps"] == 20 and sm["warmup"] == 5 and sm["scaling"] == "weak"
    cf = sm["config"]
    assert cf["rows_per_gpu"] == 12000 and cf["cols"] == 1500 and cf["parallelism"] == "row-shard x2"
    assert cf["rccl_nranks"] == 2 and "rehearsal" in cf and "row-sharded 24000x1500" in cf["workload"]
    assert sm["value"] > 0 and _close(sm["value"], 2 * 1e3 / sm["ms_per_step"], 1e-4)      # summed over the two ranks
    assert sm["solve_status"] == 0 and sm["roofline"]["frac"] > 0
    # the sharded solution against rank 0's UNSHARDED solve of the regenerated whole problem
    par = det["parity_vs_reference"]
    assert "UNSHARDED" in par["against"] and par["rel_x"] < 1e-4, par
    assert abs(par["iterations_engine"] - par["iterations_reference"]) <= 3, par
    assert _close(sm["parity"]["rel_x"], par["rel_x"])
    # rank 0's CPU leg on its own shard, extrapolated to the whole problem (t_iter ~ m)
    cb = det["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["value"] > 0 and "extrapolated" in cb, cb
    assert _close(cb["value"], cb["value_on_one_shard"] / 2) and "extrapolated" in sm["cpu_baseline"]
    # (the N = 1 extras -- handle cycles, exact setup, one-shot host call -- are not part of an N > 1 line)
    assert det["solve_status"] == 0 and det["config"]["solve_iterations"] == sm["solve_iterations"]
    print("two-process rehearsal: %.1f s wall" % wall)
    assert wall < 240.0, wall       # (about 40 s on a warm box; the bound only catches a hang-and-timeout path)
