"""time_sharded.py -- a Metropolis step of a TARGET-SHARDED model (beat_amd/models/sharded.py; SURVEY 8(e): libraries that
exceed one GPU) next to the replicated model.  All ranks share ONE GPU here (gloo: the all-gather is staged
through host memory, RCCL refuses duplicate devices), so the ranks' kernels share the device and

    sharded step (wall, all ranks)  ~  replicated step  +  all-gather  +  what the pieces cost

    python tools/time_sharded.py                                         # one rank: the replicated model, fused step
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/time_sharded.py

prints one JSON line on rank 0: ms per step, ms in the all-gather (host-staged), kernel timers, and a checksum of the chain
states after the steps (bitwise the same for every rank count and for the replicated model's step in pieces).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=int, default=16)
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--chains", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--mode", default=None, help="replicated | targets (default: targets when WORLD_SIZE > 1)")
    args = ap.parse_args()
    import beat_amd
    from beat_amd import parallel
    from beat_amd.models.sharded import TargetShardedLogp
    from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population

    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    rank = 0
    if world > 1:
        os.environ["LOCAL_RANK"] = "0"
        rank, world, _ = parallel.init("gloo")
    mode = args.mode or ("targets" if world > 1 else "replicated")
    dev = torch.device("cuda", 0)
    ctx = beat_amd.get_context(0)
    ctx.use_torch_stream()
    spec = SyntheticSpec((20,), (20,), (1.0,), T=args.targets, N=args.samples, D=3, S=25, nuc_margin=0.0, time_bounds=(0.0, 0.0))
    prob, host = build_problem(spec, device_library=True, ctx=ctx)
    lay = host["layout"]
    lo, up = lay.bounds(host["lower"], host["upper"])
    f = TargetShardedLogp(prob, ctx) if mode == "targets" else prob.compile(ctx)
    C, K = args.chains, args.steps
    Q0 = torch.from_numpy(draw_population(spec, lay, host["lower"], host["upper"], C)).to(dev)
    L0 = f.batch(Q0).clone()
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    lo_t, up_t = torch.from_numpy(lo).to(dev), torch.from_numpy(up).to(dev)
    delta = torch.randn((K + 2, C, lay.size), generator=g, device=dev, dtype=torch.float64) * (5e-4 * (up_t - lo_t))
    log_u = torch.log(torch.rand((K + 2, C), generator=g, device=dev, dtype=torch.float64))
    sc = torch.ones(C, device=dev, dtype=torch.float64)
    acc = torch.zeros(C, dtype=torch.int32, device=dev)
    gather_s = [0.0]
    if mode == "targets" and world > 1:
        inner = parallel.allgather_rows

        def timed_gather(X, n_total=None):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = inner(X, n_total)
            torch.cuda.synchronize()
            gather_s[0] += time.perf_counter() - t0
            return out
        parallel.allgather_rows = timed_gather
    for i in range(2):
        f.astep_batch(Q0, L0, delta[i], sc, lo_t, up_t, log_u[i], 2e-6, acc)
    ctx.synchronize()
    gather_s[0] = 0.0
    ctx.enable_timing(True)
    ctx.reset_timing()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2, K + 2):
        f.astep_batch(Q0, L0, delta[i], sc, lo_t, up_t, log_u[i], 2e-6, acc)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    kt = {k: ctx.kernel_time(k)[0] / K for k in ("sweep", "tables", "grouptables", "gfstack", "finish", "astep") if ctx.kernel_time(k)[1]}
    ctx.enable_timing(False)
    if rank == 0:
        print(json.dumps({"mode": mode, "ranks": world, "targets": args.targets, "samples": args.samples, "chains": C, "steps": K,
                          "ms_per_step": dt / K * 1e3, "allgather_ms_per_step": gather_s[0] / K * 1e3,
                          "ms_per_step_without_allgather": (dt - gather_s[0]) / K * 1e3,
                          "kernel_ms_per_step_this_rank": kt, "kernel": ctx.last_kernel(),
                          "state_checksum": [float(Q0.sum().item()), float(L0[:, -1].sum().item()), int(acc.sum().item())]}),
              flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
