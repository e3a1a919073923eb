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
