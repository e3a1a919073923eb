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
