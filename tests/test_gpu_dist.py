"""N > 1 on the device path without a multi-GPU node: two ranks share the one
GPU of the box (gloo transport, collectives staged through host memory) and run the device SMC and
parallel tempering of a small FFI problem; populations, betas and recorded likelihoods must equal
the 1-rank run BIT FOR BIT -- the proposal streams are keyed by the global chain id, the stage
decisions are computed identically on every rank.  On boxes with two GPUs the
same comparison runs over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nproc, backend, out):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(BEATAMD_TEST_BACKEND=backend, BEATAMD_TEST_OUT=out, OMP_NUM_THREADS="1")
    worker = os.path.join(ROOT, "tests", "_dist_gpu_worker.py")
    if nproc == 1:
        cmd = [sys.executable, worker]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("DIST_GPU_WORKER_OK") == nproc, r.stdout[-2000:]
    return np.load(out)


def _compare(a, b):
    assert int(a["nstage_checks"]) == int(b["nstage_checks"]) >= 1
    for k in ("betas", "pop", "lp", "s", "ls", "scale"):
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["lp"]).all() and a["pop"].shape[0] == 256 and a["s"].shape[0] == 128


def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    one = _run(1, "gloo", str(tmp_path / "w1.npz"))
    two = _run(2, "gloo", str(tmp_path / "w2.npz"))
    _compare(one, two)


def test_two_ranks_over_rccl_equal_one_rank(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    one = _run(1, "nccl", str(tmp_path / "w1.npz"))
    two = _run(2, "nccl", str(tmp_path / "w2.npz"))
    _compare(one, two)


def _run_shard(nproc, mode, out, backend="gloo"):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(BEATAMD_TEST_BACKEND=backend, BEATAMD_TEST_OUT=out, BEATAMD_TEST_MODE=mode, OMP_NUM_THREADS="1")
    worker = os.path.join(ROOT, "tests", "_shard_gpu_worker.py")
    if nproc == 1:
        cmd = [sys.executable, worker]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("SHARD_GPU_WORKER_OK") == nproc, r.stdout[-2000:]
    return np.load(out)


def test_target_sharded_library_equals_the_replicated_model(tmp_path):
    """SURVEY 8(e) fallback: the seismic libraries sharded by TARGET over two ranks (3 + 2 targets; two
    ranks share the one GPU, gloo) -- every rank evaluates all chains on the rows of its targets, one all-gather per
    evaluation assembles the likelihood vectors.  Against the replicated one-rank model: every dataset's logpt and
    `like` BITWISE (each target's arithmetic does not depend on where the other targets live; `like` is summed from
    the gathered vector in the fused model's order), NaN for the chain outside the library grid; a one-rank sharded run
    and the two-rank run give the same SMC populations, likelihoods and betas (identical decisions on every rank); the
    replicated SMC run (fused step kernel) agrees to 1e-12 on the betas"""
    rep = _run_shard(1, "replicated", str(tmp_path / "rep.npz"))
    one = _run_shard(1, "targets", str(tmp_path / "s1.npz"))
    two = _run_shard(2, "targets", str(tmp_path / "s2.npz"))
    for sh in (one, two):
        assert sh["LL"].shape == rep["LL"].shape == (300, 9)
        assert np.array_equal(np.isnan(sh["LL"][:, -1]), np.isnan(rep["LL"][:, -1])) and np.isnan(rep["LL"][7, -1])
        ok = ~np.isnan(rep["LL"][:, -1])
        assert np.array_equal(sh["LL"][ok], rep["LL"][ok])
        np.testing.assert_allclose(sh["LL"][ok, -1], rep["LL"][ok, -1], rtol=1e-12)
    for k in ("pop", "lp", "betas"):
        assert np.array_equal(one[k], two[k]), k
    np.testing.assert_allclose(one["betas"], rep["betas"], rtol=1e-12)
    assert np.isfinite(two["lp"]).all() and two["pop"].shape[0] == 256


def _bench(extra, env_extra, nproc_flag):
    import json
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc_flag), "--steps", "4", "--warmup", "1",
           "--targets", "4", "--samples", "256", "--no-cpu-baseline", "--no-streaming-leg", "--no-batch-leg",
           "--no-narrow-leg", "--no-variant-legs"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu_keeps_the_contract(tmp_path):
    """`bench.py --gpus 2` without a multi-GPU node -- the two ranks share the one GPU (gloo,
    collectives staged through host memory).  The driver's contract: n_gpus, global chains, value = all ranks'
    chain-steps over the slowest rank's time, a stage transition whose all-gather really had two ranks; and the
    gathered end points (proposal rows are seeded per block of 64 GLOBAL chains, the population per global chain)
    equal those of ONE rank stepping all 256 chains, bit for bit."""
    two = _bench(["--chains", "128"], {"BEATAMD_BENCH_BACKEND": "gloo"}, 2)
    one = _bench(["--chains", "256"], {}, 1)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["steps"] == one["steps"] == 4
    assert two["config"]["global_chains"] == one["config"]["global_chains"] == 256
    assert two["config"]["chains_per_gpu"] == 128 and two["scaling"] == "weak"
    for d in (one, two):
        assert abs(d["value"] - d["config"]["global_chains"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) \
            <= 1e-9 * d["value"]
        assert d["stage_transition_ms"] > 0 and d["stage_transition"]["gathered_chains"] == 256
    assert two["stage_transition"]["ranks_in_all_gather"] == 2 and two["stage_transition"]["backend"] == "gloo"
    assert two["stage_transition"]["population_checksum"] == one["stage_transition"]["population_checksum"]
    assert 0.0 < two["stage_transition"]["population_checksum"]["next_beta"] < 1.0


def test_bench_eight_ranks_on_one_gpu(tmp_path):
    """the driver's 8-rank launch line has executed before a node ever sees it -- `bench.py --gpus 8` with
    all eight ranks on the one GPU (gloo, reduced shape): seeds per block of 64 global chains, the stage all-gather over
    eight ranks, value = all ranks' chain-steps over the slowest rank's time; the gathered end points equal those of ONE
    rank stepping all 512 chains bit for bit"""
    eight = _bench(["--chains", "64"], {"BEATAMD_BENCH_BACKEND": "gloo"}, 8)
    one = _bench(["--chains", "512"], {}, 1)
    assert eight["n_gpus"] == 8 and eight["config"]["global_chains"] == one["config"]["global_chains"] == 512
    assert eight["config"]["chains_per_gpu"] == 64 and eight["scaling"] == "weak"
    assert eight["stage_transition"]["ranks_in_all_gather"] == 8 and eight["stage_transition"]["gathered_chains"] == 512
    assert eight["stage_transition"]["population_checksum"] == one["stage_transition"]["population_checksum"]
    assert abs(eight["value"] - 512 * eight["steps"] / (eight["ms_per_step"] * 1e-3 * eight["steps"])) <= 1e-9 * eight["value"]
