"""
Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests).

Replaces the reference's two mechanisms:
  * beat/parallel.py:117-282 ``paripool`` (fork pool, one chain per task) and
    :285-439 shared-memory GF library  ->  chains are the batch dimension of the kernels,
    each rank owns a contiguous block of chains and one HBM-resident library copy;
  * the implicit stage gather through trace files (sampler/smc.py:188-240
    ``select_end_points``)  ->  one all-gather of (end points, likelihood vectors) per stage.

No collective sits inside a chain step; payloads are a few MB per stage (latency bound), so
a single flat all-gather is used rather than bucketing.
"""
import os

import numpy as np


def dist_info():
    """-> (rank, world_size, local_rank) from the torchrun environment"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None):
    """Initialise the default process group if launched under torchrun (idempotent)."""
    import torch
    import torch.distributed as dist
    rank, world, local = dist_info()
    if world == 1 or dist.is_initialized():
        return rank, world, local
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def chain_block(n_chains, rank, world):
    """Contiguous block of chains owned by ``rank``: -> (start, stop).
    The reference requires n_chains / n_jobs to be whole (smc.py:419-421); here the
    remainder is spread over the first ranks."""
    base, rem = divmod(int(n_chains), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allgather_population(Q, L):
    """All ranks' end points and likelihood vectors, in global chain order.
    Q (c_local, nparams), L (c_local, nllk) torch tensors (cuda or cpu) -> (Qall, Lall)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return Q, L
    world = dist.get_world_size()   # (a single rank runs through the same collectives)
    n_local = torch.tensor([Q.shape[0]], device=Q.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    width = Q.shape[1] + L.shape[1]
    buf = torch.zeros((nmax, width), device=Q.device, dtype=Q.dtype)
    buf[:Q.shape[0], :Q.shape[1]] = Q
    buf[:Q.shape[0], Q.shape[1]:] = L
    out = torch.empty((world * nmax, width), device=Q.device, dtype=Q.dtype)
    dist.all_gather_into_tensor(out, buf)
    parts = [out[r * nmax:r * nmax + counts[r]] for r in range(world)]
    allp = torch.cat(parts, 0)
    return allp[:, :Q.shape[1]].contiguous(), allp[:, Q.shape[1]:].contiguous()


def broadcast_array(a, src=0):
    """Broadcast a small numpy array from ``src`` (stage decisions are computed identically on
    every rank; this is only used for values that come from host RNG state)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return a
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dist.broadcast(t, src)
    return t.cpu().numpy()
