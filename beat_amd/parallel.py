"""
Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests).

Replaces the reference's two mechanisms:
  * beat/parallel.py:117-282 ``paripool`` (fork pool, one chain per task) and
    :285-439 shared-memory GF library  ->  chains are the batch dimension of the kernels,
    each rank owns a contiguous block of chains and one HBM-resident library copy;
  * the implicit stage gather through trace files (sampler/smc.py:188-240
    ``select_end_points``)  ->  one all-gather of (end points, likelihood vectors) per stage;
  * the MPI star of parallel tempering (sampler/pt.py:472-704)  ->  per swap round an
    all-gather of one likelihood per replica (8 B each), the rank-identical swap decision, and
    point-to-point moves of only the rows that actually change rank.

No collective sits inside a chain step; payloads are a few MB per stage (latency bound), so
flat collectives are used rather than bucketing.
"""
import os

import numpy as np


def dist_info():
    """-> (rank, world_size, local_rank): from the initialised process group when there is one,
    else from the torchrun environment"""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", "0"))
    except ImportError:
        pass
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


def ensure_group():
    """Samplers shard chains by (rank, world): with WORLD_SIZE > 1 in the environment the process
    group must exist, otherwise the all-gathers would silently return local blocks only."""
    rank, world, local = dist_info()
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            rank, world, local = init()
        assert dist.is_initialized() and dist.get_world_size() == world
    return rank, world, local


def chain_block(n_chains, rank, world):
    """Contiguous block of chains owned by ``rank``: -> (start, stop).
    The reference requires n_chains / n_jobs to be whole (smc.py:419-421); here the
    remainder is spread over the first ranks."""
    base, rem = divmod(int(n_chains), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def _staged(X):
    """The transport buffer of a collective.  RCCL ("nccl") moves device tensors directly.  A gloo group
    fed with device tensors -- two ranks sharing ONE GPU in tests/test_gpu_dist.py, where RCCL refuses
    duplicate devices -- goes through host memory: -> (tensor to hand to torch.distributed,
    function that brings a result back to X's device)."""
    import torch.distributed as dist
    if X.is_cuda and dist.get_backend() == "gloo":
        dev = X.device
        return X.cpu(), (lambda t: t.to(dev))
    return X, (lambda t: t)


def allgather_rows(X, n_total=None):
    """rows of all ranks in rank order; X (n_local, width) -> (n_total, width).  Blocks may differ
    in length.  ``n_total`` (the global row count, when the caller knows it and the blocks are those of
    ``chain_block``): the block lengths follow from it and the exchange of the counts -- a second collective
    and a host synchronisation per call -- is skipped (one likelihood all-gather per PT exchange round)."""
    import torch
    import torch.distributed as dist
    if not _active():
        return X
    world = dist.get_world_size()   # (a single rank runs through the same collectives)
    Xt, back = _staged(X)
    if n_total is not None:
        counts = [b - a for a, b in (chain_block(n_total, r, world) for r in range(world))]
        if counts[dist.get_rank()] != Xt.shape[0]:
            raise ValueError("allgather_rows: this rank holds %d rows, chain_block(%d) gives it %d"
                             % (Xt.shape[0], n_total, counts[dist.get_rank()]))
    else:
        n_local = torch.tensor([Xt.shape[0]], device=Xt.device, dtype=torch.int64)
        counts = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(counts, n_local)
        counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    buf = torch.zeros((nmax,) + tuple(Xt.shape[1:]), device=Xt.device, dtype=Xt.dtype)
    buf[:Xt.shape[0]] = Xt
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        out = torch.cat(parts, 0)
    else:
        out = torch.empty((world * nmax,) + tuple(Xt.shape[1:]), device=Xt.device, dtype=Xt.dtype)
        dist.all_gather_into_tensor(out, buf)
    if not all(c == nmax for c in counts):
        out = torch.cat([out[r * nmax:r * nmax + counts[r]] for r in range(world)], 0)
    return back(out)


def allgather_population(Q, L, n_total=None):
    """All ranks' end points and likelihood vectors, in global chain order.
    Q (c_local, nparams), L (c_local, nllk) torch tensors (cuda or cpu) -> (Qall, Lall)."""
    import torch
    if not _active():
        return Q, L
    allp = allgather_rows(torch.cat([Q, L], 1), n_total)
    return allp[:, :Q.shape[1]].contiguous(), allp[:, Q.shape[1]:].contiguous()


def exchange_rows(X, perm, n_total, gather=None):
    """Apply a global row permutation to a population sharded in contiguous blocks:
    new_global[i] = old_global[perm[i]].  X (n_local, width) is this rank's block; perm is the
    same (n_total,) integer array on every rank.  Rows whose source is local are moved on the
    device (``gather(src, idx)``, default torch indexing); only rows that change rank travel,
    point to point (parallel tempering swaps adjacent temperatures, so that is a handful of rows
    at block boundaries -- pt.py:573-633 ships every state through the master instead)."""
    import torch
    import torch.distributed as dist
    perm = np.asarray(perm, dtype=np.int64)
    if _active():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    blocks = [chain_block(n_total, r, world) for r in range(world)]
    start, stop = blocks[rank]
    owner = np.empty(n_total, dtype=np.int64)
    for r, (a, b) in enumerate(blocks):
        owner[a:b] = r
    src = perm[start:stop]
    local = owner[src] == rank
    idx = np.where(local, src - start, 0).astype(np.int32)
    idx_t = torch.from_numpy(idx).to(X.device)
    out = gather(X, idx_t) if gather is not None else X[idx_t.long()].contiguous()
    if world == 1 or local.all() and all((owner[perm[a:b]] == r).all() for r, (a, b) in enumerate(blocks)):
        return out
    ops, recv = [], []
    for r in range(world):
        if r == rank:
            continue
        a, b = blocks[r]
        # rows rank r needs from this rank, in the order of r's destination rows
        send_src = perm[a:b][owner[perm[a:b]] == rank] - start
        if send_src.size:
            sbuf = _staged(X[torch.from_numpy(send_src).to(X.device)].contiguous())[0]
            ops.append(dist.P2POp(dist.isend, sbuf, r))
        # rows this rank needs from rank r
        need = np.nonzero(owner[src] == r)[0]
        if need.size:
            tdev = _staged(X[:0])[0].device
            rbuf = torch.empty((need.size,) + tuple(X.shape[1:]), device=tdev, dtype=X.dtype)
            ops.append(dist.P2POp(dist.irecv, rbuf, r))
            recv.append((need, rbuf))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for need, rbuf in recv:
        out[torch.from_numpy(need).to(X.device)] = rbuf.to(X.device)
    return out


def assert_same_on_all_ranks(what, *tensors):
    """Debug aid (BEATAMD_CHECK_RANKS=1): the stage decisions are computed redundantly on every
    rank from the gathered population and MUST be bit-identical -- a silent divergence would let the
    ranks resample different parents.  All-gathers the raw bytes of the given tensors and raises if any
    rank's differ from rank 0's."""
    import torch
    if not _active():
        return
    for i, t in enumerate(tensors):
        t = torch.as_tensor(t)
        flat = t.detach().contiguous().reshape(-1)
        if flat.dtype != torch.float64:
            flat = flat.to(torch.float64) if flat.dtype in (torch.int32, torch.int64, torch.float32) else flat.double()
        allr = allgather_rows(flat[None])
        if not bool((allr.view(torch.int64) == allr.view(torch.int64)[0:1]).all()):
            bad = [r for r in range(allr.shape[0]) if not torch.equal(allr[r].view(torch.int64), allr[0].view(torch.int64))]
            raise RuntimeError("%s: tensor %d differs between rank 0 and rank(s) %s" % (what, i, bad))


def broadcast_array(a, src=0):
    """Broadcast a small numpy array from ``src`` (stage decisions are computed identically on
    every rank; this is only used for values that come from host RNG state)."""
    import torch
    import torch.distributed as dist
    if not _active() or dist.get_world_size() == 1:
        return a
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"   # (a few bytes: always through the transport's memory)
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dist.broadcast(t, src)
    return t.cpu().numpy()
