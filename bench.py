#!/usr/bin/env python
"""
bench.py -- SMC chain-steps/s of the FFI seismic hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2], SURVEY.md 8(d) "config 3"): one 20x20-patch subfault
(P=400, 1 km), T=64 targets, N=4096 samples, library (64,400,3,25,4096) float64 = 62.9 GB
resident in HBM, N(0,1) values generated on device (synthetic), covariance sigma^2 I
(``--covariance toeplitz`` for the dense-W variant), nearest-neighbour interpolation
(``--interp multilinear`` for the 4-row blend).  Chain population exactly as SURVEY 8(d):
slips U(0,5), durations U(0.5,1.5), velocities U(2.5,4.0), nucleation anywhere on the fault
(U over the 20x20 patches), origin time 0, chain c from default_rng(1000 + c).  (Round 1 confined
the hypocentre to the central 7.5 km and is kept as ``--prior narrow`` for comparison only.)

One "step" = one batched Metropolis.astep (reference beat/sampler/metropolis.py:276-422)
for ``--chains`` chains per GPU: proposal, prior-box check, forward model
(fast sweep -> start times -> GF stacking -> residual -> sum logp), tempered MH accept.
All inputs (library, data, weights, chain states, proposal rows, uniforms) are resident in
HBM when the timed region starts.

    python bench.py                        # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Chains are sharded over ranks (weak scaling: fixed chains per GPU); there is no collective
inside a step -- the SMC stage transition (all-gather of end points, once per ~100-400
steps) is timed separately and reported as ``stage_transition_ms``.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PROFILE_ROUND = "r6"   # prefix of the counter summaries under profiles/ (tools/run_gpu.sh profile6: one lease)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
LDS_PEAK_GBS = 256 * 256 * 2.4   # 256 CUs x 256 B/clk (ds_read_b64, MI355X_MICROARCH.md LDS table) x 2.4 GHz
FP64_VALU_PEAK_TFLOPS = 78.6

LINE_CAP = 8192      # the driver keeps an 8 KB tail of stdout: the LAST line must fit in it with room to spare


def _r(x, sig=5):
    """float -> `sig` significant digits (ints, None, strings pass)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (sig, x))


def _leg_summaries(out):
    """one {cps, ms, k, frac} entry per timed leg of the full result object, keyed by its path"""
    legs = {}

    def visit(path, d):
        cps = d.get("chain_steps_per_s", d.get("chain_steps_per_s_sampling_only"))
        if cps is not None and path:
            e = {"cps": _r(float(cps), 4)}
            ms = d.get("ms_per_step", d.get("ms_per_step_incl_exchange", d.get("gfstack_avg_launch_ms", d.get("avg_launch_ms"))))
            if ms is not None:
                e["ms"] = _r(float(ms), 4)
            roof = d.get("roofline") if isinstance(d.get("roofline"), dict) else (d if "frac" in d else None)
            k = d.get("kernel") or (roof or {}).get("kernel")
            if k:
                e["k"] = k.replace(" ", "")
            if roof is not None and roof.get("frac") is not None:
                e["frac"] = _r(float(roof["frac"]), 3)
                if roof.get("traffic") is not None:
                    e["pmc"] = 1          # (frac from counter bytes of a committed summary of the same command)
            legs[path] = e
        for k_, v in d.items():
            if isinstance(v, dict) and k_ not in ("roofline", "config", "cpu_baseline", "stage_transition",
                                                  "roofline_quadform", "plan", "split_s"):
                visit((path + "." if path else "") + k_.replace("_leg", ""), v)

    visit("", {k: v for k, v in out.items() if isinstance(v, dict)})
    return legs


def compact_line(out, full_path=None):
    """The ONE stdout line of a run: the contract fields, `roofline` (counter-based fraction where a PMC summary of the same
    command is committed), `roofline_streaming` (SURVEY 8(d)'s algorithmic bytes, no cross-chain reuse), `cpu_baseline` and a
    one-entry-per-leg summary.  Everything else (notes, plans, per-kernel splits) goes to the full object on disk.
    Guaranteed < LINE_CAP bytes: legs are dropped from the end, then strings cut, before the cap is ever exceeded."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "repeats")
    line = {k: out[k] for k in keep if k in out}      # (full precision: value = chains x steps / time is checked)
    cfg = out.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "chains_per_gpu", "global_chains", "parallelism") if k in cfg}
    if cfg.get("env_knobs"):
        line["config"]["env_knobs"] = cfg["env_knobs"]
    rk = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
          "hbm_counter_frac", "hbm_required_bytes_per_launch", "hbm_frac_required_bytes", "fp64_valu_frac", "lds_frac",
          "frac_basis", "traffic_source", "launches", "chains")
    for name in ("roofline", "roofline_streaming"):
        if isinstance(out.get(name), dict):
            line[name] = {k: _r(out[name][k], 5) for k in rk if k in out[name]}
    if isinstance(out.get("cpu_baseline"), dict):
        ck = ("value", "unit", "cores", "kind", "sample", "value_1core", "value_1core_numpy_reference_path",
              "cgroup_cpu_quota_cores", "cpus_visible")
        line["cpu_baseline"] = {k: _r(out["cpu_baseline"][k], 5) for k in ck if k in out["cpu_baseline"]}
    for k in ("speedup_vs_cpu_baseline", "stage_transition_ms", "in_box_fraction", "setup_s"):
        if k in out:
            line[k] = _r(out[k], 5)
    if isinstance(out.get("stage_transition"), dict):
        line["stage_transition"] = out["stage_transition"]       # (ranks in the all-gather, population checksum: ~200 bytes)
    if "kernel_ms_per_step" in out:
        line["kernel_ms_per_step"] = {k: _r(v, 4) for k, v in out["kernel_ms_per_step"].items()}
    legs = _leg_summaries(out)
    legs.pop("roofline_streaming", None)
    line["legs"] = legs
    if full_path:
        line["full"] = os.path.basename(full_path)
    txt = json.dumps(line, separators=(",", ":"), allow_nan=False)
    names = list(legs)
    while len(txt) >= LINE_CAP - 512 and names:       # never reached by the shipped set of legs (tests/test_bench_contract.py)
        legs.pop(names.pop())
        line["legs_truncated"] = True
        txt = json.dumps(line, separators=(",", ":"), allow_nan=False)
    if len(txt) >= LINE_CAP - 512:
        line["config"]["workload"] = line["config"].get("workload", "")[:200]
        if "cpu_baseline" in line:
            line["cpu_baseline"]["sample"] = str(line["cpu_baseline"].get("sample", ""))[:120]
        txt = json.dumps(line, separators=(",", ":"), allow_nan=False)
    assert len(txt) < LINE_CAP, len(txt)
    return txt


def algorithmic_bytes_per_chain_step(spec, nvar=1):
    """gathered rows + index/slip tables + the data read of the fused
    residual (replaces the synthetics write of the unfused form, same size)."""
    T, P, N = spec.T, spec.P, spec.N
    rows = 4 if spec.interpolation == "multilinear" else 1
    gathered = T * P * N * 8 * rows * nvar
    tables = T * P * (4 * rows + (8 * rows if rows == 4 else 0)) + P * 8 * nvar
    data = T * N * 8
    return gathered + tables + data


def _numa_nodes(cpus):
    """-> [(node id, [cpus of this process on the node])] from sysfs; one pseudo node when sysfs has none"""
    import glob
    nodes = []
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"), key=lambda x: int(x.rsplit("node", 1)[1])):
        try:
            lst = open(os.path.join(d, "cpulist")).read().strip()
        except OSError:
            continue
        mine = []
        for part in lst.split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            mine += [c for c in range(int(a), int(b or a) + 1) if c in cpus]
        if mine:
            nodes.append((int(d.rsplit("node", 1)[1]), mine))
    return nodes or [(0, sorted(cpus))]


def _make_cpu_eval(G, spec, data, w, slog, ml):
    from oracle import oracle as orc  # noqa: F401  (cpu_baseline leg only)
    P = spec.P
    lib_cfg = dict(dur_min=spec.du_min, dur_dt=spec.du_dt, st_min=spec.st_min, st_dt=spec.st_dt)
    fault = dict(ndip=spec.n_patch_dip, nstrike=spec.n_patch_strike, patch_size=spec.patch_size)

    def one(seed):
        r = np.random.default_rng(seed)
        params = dict(slips=r.uniform(0, 5, (1, P)), durations=r.uniform(0.55, 0.7, P) if not ml else r.uniform(0.55, 0.95, P),
                      velocities=r.uniform(*spec.vel_bounds, P),
                      nuc_strike=[r.uniform(0, 19.49)], nuc_dip=[r.uniform(0, 19.49)], time=[0.0])
        orc.ffi_seismic_forward([G], lib_cfg, fault, params, data, w, slog, 0.0,
                                interpolation=spec.interpolation, return_synthetics=False)
    return one


def _cpu_worker_main(i, cpu, node_slot, first_of_node, shape, bufs, G_parent, ctx_args, barrier, budget, out_q):
    """one forked worker = one chain at a time on one core (iter_parallel_chains, beat/sampler/base.py:428-595),
    pinned; it reads the library copy of ITS NUMA node, first-touched by the node's first worker"""
    try:
        os.sched_setaffinity(0, {cpu})
    except OSError:
        pass
    G = np.frombuffer(bufs[node_slot], dtype=np.float64).reshape(shape)
    if first_of_node:
        step = max(1, shape[0] // 8)
        for a in range(0, shape[0], step):      # first touch -> the pages land on this worker's node
            G[a:a + step] = G_parent[a:a + step]
    barrier.wait()
    one = _make_cpu_eval(G, *ctx_args)
    one(10 ** 6 + i)                            # warm (page tables)
    barrier.wait()
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < budget:
        one(1000 * i + k)
        k += 1
    out_q.put((i, k, time.perf_counter() - t0))


def cpu_baseline(spec, seconds=14.0):
    """Reference-equivalent CPU path on the host cores of this box, on a bounded sample: T_sub of the T targets with the
    full P x N gather per target, scaled to a whole chain-step.
      * the C restatement (oracle/beat_oracle.c bo_ffi_seismic_forward, pinned to the reference) on one core and on all
        cores: one forked, PINNED worker per core, one chain at a time each (iter_parallel_chains,
        beat/sampler/base.py:428-595), the library shared read-only -- ONE COPY PER NUMA NODE, first-touched by a worker
        of that node (round 4 shared the parent's copy: every page on one node, 256 workers scaled x3.9);
      * the reference's numpy stack_all arithmetic (fancy-index gather + product + einsum, beat/ffi/base.py:651-661),
        which is what SURVEY 8(d) names, timed on one core beside it."""
    import mmap
    from multiprocessing import get_context

    from oracle import oracle as orc  # noqa: F401  (cpu_baseline leg only)

    cpus = set(os.sched_getaffinity(0))
    nodes = _numa_nodes(cpus)
    # the cores this container may actually USE: its cgroup CPU quota (cpu.max = "<quota> <period>") can be far below the
    # CPUs it may run on -- the round-5 box shows 256 CPUs and a quota of 16: more workers than that only add throttling
    # (tools/host_scaling_probe.py: compute-bound workers scale x9.5 to 16 workers and not beyond; 256 workers streaming
    # private arrays reach a quarter of what 16 do).  One worker per quota core, spread over the NUMA nodes.
    quota = None
    try:
        q_, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q_ == "max" else max(1, int(round(float(q_) / float(p_))))
    except (OSError, ValueError):
        pass
    if quota is not None and quota < len(cpus):
        per = max(1, quota // len(nodes))
        nodes = [(nid, ncp[:per]) for nid, ncp in nodes]
    ncores = sum(len(ncp) for _, ncp in nodes)
    P, N, S = spec.P, spec.N, spec.S
    ml = spec.interpolation == "multilinear"
    D_cpu = 2 if ml else 1  # D reduced: same gathered volume per step, fits host RAM
    # targets per evaluation: 16 of 64 when the host has room for a copy per node (+ the parent's), else fewer
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except ImportError:
        avail = 16e9
    T_sub = 16
    while T_sub > 2 and (len(nodes) + 1.5) * T_sub * P * D_cpu * S * N * 8 > 0.5 * avail:
        T_sub //= 2
    T_sub = min(T_sub, spec.T)
    rng = np.random.default_rng(spec.seed)
    shape = (T_sub, P, D_cpu, S, N)
    G = np.empty(shape)
    base = rng.standard_normal(shape[1:])       # (one target's rows drawn, the others scaled copies: the gather and the
    for t in range(T_sub):                      #  FMAs cost the same, 5 GB of normals would take the leg's whole budget)
        np.multiply(base, 1.0 + 1e-3 * t, out=G[t])
    del base
    data = rng.standard_normal((T_sub, N))
    w = np.full(T_sub, 2.0)
    slog = np.zeros(T_sub)
    ctx_args = (spec, data, w, slog, ml)
    one = _make_cpu_eval(G, *ctx_args)
    one(0)
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < seconds / 4:
        one(n1)
        n1 += 1
    t1 = (time.perf_counter() - t0) / n1
    rate1 = (T_sub / spec.T) / t1  # full chain-steps/s on one core
    row_bytes_eval = float(T_sub) * P * N * 8 * (4 if ml else 1)
    # ---- the reference's numpy path on one core (same rows, same volume): gather copy, product, sum over patches
    didx = np.zeros(P, dtype=np.int16)
    tidx = np.arange(T_sub)[:, None]
    pidx = np.arange(P)
    r_ = np.random.default_rng(1)
    n_np, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds / 8 or n_np == 0:
        sidx = r_.integers(0, S, (T_sub, P)).astype(np.int16)
        slips = r_.uniform(0, 5, P)
        cut = G[tidx, pidx, didx, sidx, :]                       # base.py:651-656 (nn; multilinear does this four times)
        out_np = np.einsum("ijk->ik", cut * slips[None, :, None])   # base.py:658-661
        n_np += 1
    t_np = (time.perf_counter() - t0) / n_np * (4 if ml else 1)
    rate_numpy = (T_sub / spec.T) / t_np
    del cut, out_np
    # ---- all cores
    rate_all, nsteps, dt_all, per_node = rate1, n1, t1 * n1, {nodes[0][0]: 1}
    if ncores > 1:
        budget = seconds / 2
        mp = get_context("fork")
        bufs = [mmap.mmap(-1, G.nbytes) for _ in nodes]          # anonymous shared mappings, untouched until a worker fills them
        barrier = mp.Barrier(ncores)
        out_q = mp.Queue()
        procs, i = [], 0
        for slot, (node, ncpus) in enumerate(nodes):
            for k, cpu in enumerate(ncpus):
                pr = mp.Process(target=_cpu_worker_main, args=(i, cpu, slot, k == 0, shape, bufs, G, ctx_args, barrier,
                                                               budget, out_q))
                pr.start()
                procs.append(pr)
                i += 1
        res = [out_q.get(timeout=600) for _ in procs]
        for pr in procs:
            pr.join()
        nsteps = int(sum(r[1] for r in res))
        dt_all = max(r[2] for r in res)
        rate_all = nsteps * (T_sub / spec.T) / dt_all
        per_node = {str(node): len(ncpus) for node, ncpus in nodes}
        for b in bufs:
            b.close()
    phys = None
    try:   # physical cores: distinct (package, core) pairs of the CPUs this process may use
        ids = set()
        for c in sorted(c_ for _, ncp in nodes for c_ in ncp):
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            ids.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        phys = len(ids)
    except OSError:
        pass
    return dict(value=rate_all, unit="chain-steps/s", cores=ncores, cores_physical=phys, cpus_visible=len(cpus),
                cgroup_cpu_quota_cores=quota, kind="port",
                cores_note="one pinned worker per core of the container's cgroup CPU quota (cpu.max); the CPUs it may run ON "
                           "are more, but workers beyond the quota only get throttled (tools/host_scaling_probe.py, "
                           "profiles/r5_host_probe.json: compute-bound workers stop scaling at the quota)",
                extrapolated_from="%d of %d targets per evaluation, scaled to a full chain-step" % (T_sub, spec.T),
                value_1core=rate1, scaling_all_cores_vs_1core=rate_all / rate1,
                value_1core_numpy_reference_path=rate_numpy,
                numpy_note="the reference's numpy stack_all arithmetic (fancy-index gather, product, einsum; beat/ffi/base.py:651-661) "
                           "on the same sample, one core: what SURVEY 8(d) names; the C port is %.1f x that per core" % (rate1 / rate_numpy),
                host_gather_GBs_all_cores=nsteps * row_bytes_eval / dt_all / 1e9,
                host_gather_GBs_1core=row_bytes_eval / t1 / 1e9,
                numa_nodes=len(nodes), workers_per_numa_node=per_node,
                library_copies="one per NUMA node, first-touched by a pinned worker of that node",
                sample="%d of %d targets (full %dx%d gather per target), %d sample evaluations, "
                       "oracle/beat_oracle.c (C restatement of the reference numpy/C path), "
                       "%d forked pinned workers x 1 thread" % (T_sub, spec.T, P, N, nsteps, ncores))


def make_spec(args):
    from beat_amd.synthetic import SyntheticSpec
    prior = dict(nuc_margin=0.0, time_bounds=(0.0, 0.0)) if args.prior == "survey" \
        else dict(nuc_margin=6.0, time_bounds=(0.0, 0.5))
    return SyntheticSpec((20,), (20,), (1.0,), T=args.targets, N=args.samples, D=args.ndurations,
                         S=args.nstarttimes, du_min=args.duration_min, du_dt=args.duration_sampling,
                         covariance=args.covariance, interpolation=args.interp, **prior)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed regions of --steps steps each; `value` is their median")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "bench_full.json"),
                    help="where the complete result object (all legs, notes) is written; stdout gets the compact line")
    ap.add_argument("--chains", type=int, default=512, help="chains per GPU per step")
    ap.add_argument("--interp", default="nearest_neighbor",
                    choices=["nearest_neighbor", "multilinear"])
    ap.add_argument("--covariance", default="scalar", choices=["scalar", "toeplitz"])
    ap.add_argument("--targets", type=int, default=64)
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--nstarttimes", type=int, default=25)
    ap.add_argument("--ndurations", type=int, default=3)
    ap.add_argument("--duration-min", type=float, default=0.5, help="first node of the library's duration axis [s]")
    ap.add_argument("--duration-sampling", type=float, default=0.5, help="spacing of the duration axis [s]; "
                    "`--samples 512 --ndurations 17 --nstarttimes 41 --duration-min 0 --duration-sampling 0.25` makes the "
                    "tutorial-grid library of realistic_grid_leg the main workload (rocprof runs)")
    ap.add_argument("--prior", default="survey", choices=["survey", "narrow"],
                    help="survey: SURVEY 8(d) population (default); narrow: round 1's confined hypocentre")
    ap.add_argument("--step-scale", type=float, default=5e-4,
                    help="proposal standard deviation as a fraction of the prior span of every parameter")
    ap.add_argument("--prewhiten", action="store_true",
                    help="toeplitz only: whiten the library once (W.G), no dense W.r per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streaming-leg", action="store_true",
                    help="skip the reuse-free streaming-kernel leg (roofline_streaming)")
    ap.add_argument("--no-batch-leg", action="store_true", help="skip the labelled 2048-chain batch leg")
    ap.add_argument("--no-narrow-leg", action="store_true",
                    help="skip the labelled round-1 narrow-prior leg")
    ap.add_argument("--no-variant-legs", action="store_true",
                    help="skip the labelled legs of the other configurations: multilinear (the reference's "
                         "default interpolation), dense Toeplitz covariance, pre-whitened library, parallel "
                         "tempering, geometry mode")
    ap.add_argument("--variant-legs", default="multilinear,toeplitz,default_config,stage_update,smc,pt,prewhitened,geometry,fp32,config4,realistic_grid,sharded",
                    help="which of the labelled configuration legs to run (comma separated)")
    ap.add_argument("--profiles-dir", default=os.path.join(ROOT, "profiles"),
                    help="where the committed rocprofv3 counter summaries (%s_*_summary.json) of the legs' commands are read from"
                         % PROFILE_ROUND)
    ap.add_argument("--pmc-summary", default=None,
                    help="rocprofv3 PMC summary (tools/run_profile.sh + tools/summarize_rocpd.py) of THIS "
                         "command; supplies roofline.traffic when its configuration matches")
    ap.add_argument("--gf-order", type=int, default=None,
                    help="k_gfstack block order: 0 (chain,target,tile) 1 (target,chain,tile)")
    args = ap.parse_args()

    if args.gf_order is not None:
        os.environ["BEATAMD_GF_ORDER"] = str(args.gf_order)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one per GPU)
        # through torch.distributed.run and hand its output through; under torchrun this is skipped
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    # BEATAMD_BENCH_BACKEND=gloo (tests/test_gpu_dist.py): every rank uses cuda:0 -- N ranks share ONE GPU and the
    # collectives are staged through host memory (beat_amd.parallel._staged; RCCL refuses duplicate devices).
    # The contract (n_gpus, global chains, value = all ranks' chain-steps / max time, a stage transition whose
    # all-gather really had N ranks) is exercised without a multi-GPU node; the numbers are not a scaling curve.
    backend = os.environ.get("BEATAMD_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # BEATAMD_BENCH_FORCE_DIST=1: exercise the RCCL path with a single rank (1-GPU boxes)
    use_dist = world > 1 or bool(os.environ.get("BEATAMD_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    import beat_amd
    from beat_amd.sampler import SMC
    from beat_amd.synthetic import build_problem, draw_population

    dev = torch.device("cuda", local_rank)
    ctx = beat_amd.get_context(local_rank)
    ctx.use_torch_stream()
    env_knobs = sorted(k for k in os.environ if k.startswith("BEATAMD_G"))

    spec = make_spec(args)
    t_build = time.perf_counter()
    prob, host = build_problem(spec, device_library=True, ctx=ctx)
    f = prob.compile(ctx, prewhiten="inplace" if args.prewhiten else False)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    B, K, W = args.chains, args.steps, args.warmup
    lay = host["layout"]
    lo, up = lay.bounds(host["lower"], host["upper"])

    PBLK = 64   # proposal rows are seeded per block of 64 GLOBAL chains: the same rows whatever the sharding

    def seeded_blocks(draw, n_chains, first_chain, tail):
        """(n_steps + n_warm, n_chains) + tail draws, chains [64 j, 64 j + 64) of the global population from
        generator seed 4242 + j"""
        assert first_chain % PBLK == 0, "chains per GPU must be a multiple of %d" % PBLK
        gen = torch.Generator(device=dev)
        parts = []
        for j in range((n_chains + PBLK - 1) // PBLK):
            gen.manual_seed(4242 + first_chain // PBLK + j)
            parts.append(draw((min(PBLK, n_chains - j * PBLK),) + tail, gen))
        return torch.cat(parts, 0)

    def run_leg(spec_leg, f_leg, n_chains, n_steps, n_warm, seed_offset, beta=2e-6, repeats=1):
        """`repeats` timed regions of n_steps astep batches of n_chains chains each (every region bracketed by a barrier
        and a device synchronisation on both sides); everything resident in HBM beforehand.
        -> dict(dt = median region, dts, kernel times over all regions, in-box fraction, acceptance)"""
        box = host_of[spec_leg]
        lay_l = box.get("layout", lay)   # the legs of the main library share its parameter layout; only the prior box differs
        Q0 = torch.from_numpy(draw_population(spec_leg, lay_l, box["lower"], box["upper"], n_chains,
                                              seed_offset=seed_offset)).to(dev)
        lo_h, up_h = lay_l.bounds(box["lower"], box["upper"])
        lo_s, up_s = torch.from_numpy(lo_h).to(dev), torch.from_numpy(up_h).to(dev)
        L0 = f_leg.batch(Q0)
        # proposal rows: proposal_samples_array[stage_sample] of every chain (metropolis.py:289-313)
        first = seed_offset - 1000      # (the global index of this leg's first chain)
        nsw = n_steps * repeats + n_warm
        delta = seeded_blocks(lambda shp, g: torch.randn(shp, generator=g, device=dev, dtype=torch.float64),
                              n_chains, first, (nsw, lay_l.size)).permute(1, 0, 2).contiguous()
        delta = delta * (args.step_scale * (up_s - lo_s))
        log_u = torch.log(seeded_blocks(lambda shp, g: torch.rand(shp, generator=g, device=dev, dtype=torch.float64),
                                        n_chains, first, (nsw,)).permute(1, 0).contiguous())
        scaling = torch.ones(n_chains, device=dev, dtype=torch.float64)
        accepted = torch.zeros(n_chains, device=dev, dtype=torch.int32)
        # fraction of proposals inside the prior box (the reference evaluates the forward model
        # only for those, metropolis.py:335-385; the batch evaluates every chain)
        q = Q0[None] + delta[n_warm:]
        in_box = float(((q >= lo_s) & (q <= up_s)).all(-1).double().mean().item())
        del q
        ctx.synchronize()

        def step(i):
            f_leg.astep_batch(Q0, L0, delta[i], scaling, lo_s, up_s, log_u[i], beta, accepted)

        for i in range(n_warm):
            step(i)
        ctx.synchronize()  # also surfaces an out-of-library index as an exception
        ctx.enable_timing(True)
        ctx.reset_timing()
        n_acc = 0
        dts = []
        for rep in range(repeats):
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_warm + rep * n_steps, n_warm + (rep + 1) * n_steps):
                step(i)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            dts.append(time.perf_counter() - t0)
        ctx.synchronize()
        n_acc = int(accepted.sum().item())
        # (device-event sums over all regions, divided by the launches they cover: per-launch averages)
        times = {k: ctx.kernel_time(k) for k in ("sweep", "tables", "grouptables", "gfstack", "quadform",
                                                 "finish", "astep")}
        ctx.enable_timing(False)
        return dict(dt=sorted(dts)[len(dts) // 2], dts=dts, times=times, in_box=in_box,
                    accept_last=n_acc / float(n_chains), steps_timed=n_steps * repeats,
                    kernel=ctx.last_kernel(), stats=ctx.gf_group_stats(), Q=Q0, L=L0)

    host_of = {spec: host}
    # `value`: the MEDIAN of `--repeats` timed regions of exactly K steps each (max over ranks per region)
    R = max(1, args.repeats)
    main_leg = run_leg(spec, f, B, K, W, seed_offset=1000 + rank * B, repeats=R)
    tmax = torch.tensor(main_leg["dts"], device="cpu" if backend == "gloo" else dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    region_s = [float(x) for x in tmax.tolist()]
    dt_max = sorted(region_s)[len(region_s) // 2]

    # ---- SMC stage transition on the end points of the timed steps (select_end_points ->
    # calc_beta -> proposal factor -> resample -> restart points, smc.py:133-324): all-gather over
    # the ranks + the device kernels; outside the timed steps (it happens once per ~100-400 steps)
    smc = SMC(f, lo, up, n_chains=world * B, device=dev, random_seed=7)
    for rep in range(2):   # first pass allocates
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        smc.beta = 0.0
        smc.select_end_points(main_leg["Q"], main_leg["L"])
        smc.transition()
        Qn, Ln = smc.restart_points()
        torch.cuda.synchronize()
        stage_ms = (time.perf_counter() - t1) * 1e3
    assert Qn.shape == main_leg["Q"].shape and smc.Q_all.shape[0] == world * B
    # the gathered end points of all ranks (identical on every rank): sums as a sharding-independent fingerprint
    pop_sum = (float(smc.Q_all.sum().item()), float(smc.L_all[:, -1].sum().item()), float(smc.beta))

    def stack_roofline(spec_leg, leg, n_chains):
        """roofline of the stacking kernel of one leg: every candidate bound with its fraction, the
        largest named `bound` (none is ever above 1); SURVEY 8(d)'s independent-chain byte count is
        kept as a figure (`algorithmic_equiv_GBs`), not a fraction, for the chain-shared kernels"""
        gf_ms, gf_n = leg["times"]["gfstack"]
        nvar = len(spec_leg.slip_varnames)
        alg = algorithmic_bytes_per_chain_step(spec_leg, nvar) * n_chains  # per launch
        rows_per_patch = 4 if spec_leg.interpolation == "multilinear" else 1
        avg_ms = gf_ms / max(gf_n, 1)
        st = leg["stats"]
        shared = st["row_bytes"] > 0
        ml_runs = leg["kernel"].startswith("k_gfstack_runs")
        # bytes the kernel has to move from HBM: every distinct row of every (group, target, patch) and slip variable once
        # (chain-shared kernels; with row passes a row that two passes need is counted twice) or every chain's rows
        # (streaming kernel), + tables
        tables = n_chains * spec_leg.T * spec_leg.P * 4 * rows_per_patch * 2 + spec_leg.T * spec_leg.N * 8
        need_bytes = (st["row_bytes"] if shared else
                      float(n_chains) * spec_leg.T * spec_leg.P * spec_leg.N * 8 * rows_per_patch * nvar) + tables
        # LDS operands: one 8-byte operand per FMA and lane for the lane <-> chain kernels (k_gfstack_runs reads a cell's
        # rows once per run of chains sharing it: its LDS operand bytes depend on the population -> not modelled here)
        lds_bytes = 0.0 if ml_runs else float(n_chains) * spec_leg.T * spec_leg.P * spec_leg.N * 8 * rows_per_patch * nvar
        lds_floor_ms = lds_bytes / (LDS_PEAK_GBS * 1e9) * 1e3
        flops = 2.0 * n_chains * spec_leg.T * spec_leg.P * spec_leg.N * rows_per_patch * nvar
        t = avg_ms * 1e-3
        hbm_frac = need_bytes / t / 1e9 / HBM_PEAK_GBS if gf_n else 0.0
        lds_frac = lds_floor_ms / avg_ms if (gf_n and shared) else 0.0
        valu_frac = flops / t / 1e12 / FP64_VALU_PEAK_TFLOPS if gf_n else 0.0
        # `bound`/`frac`: HBM bytes the kernel has to move (distinct rows, = what the PMC counters read: attach_traffic
        # replaces the figure by the counter bytes when a summary of the same command is committed) over the peak; the LDS-gather
        # and FP64 fractions and SURVEY 8(d)'s independent-chain byte count ride along as figures
        roof = {
            "bound": "hbm",
            "kernel": leg["kernel"],
            "achieved": need_bytes / t / 1e9 if gf_n else 0.0,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": hbm_frac,
            "frac_basis": "required bytes (distinct rows + tables) / HIP-event launch time",
            # HBM bytes per launch from the PMC counters of the same command (profiles/), else null
            "traffic": None,
            "hbm_frac_required_bytes": hbm_frac,
            "hbm_required_bytes_per_launch": need_bytes,
            "hbm_required_bytes_note": "distinct row segments every chain group stages (+ tables); with several groups per launch the "
                                       "groups of a (target, tile) share an XCD and rows they have in common come from its L2, so the "
                                       "HBM counters can read LESS than this (the fraction is then an upper bound of the HBM share)",
            "lds_frac": lds_frac,
            "lds_gather_bytes_per_launch": lds_bytes,
            "lds_floor_ms": lds_floor_ms,
            "fp64_valu_frac": valu_frac,
            "fp64_flops_per_launch": flops,
            "algorithmic_bytes_per_launch": alg,
            "algorithmic_equiv_GBs": alg / t / 1e9 if gf_n else 0.0,
            "avg_launch_ms": avg_ms,
            "launches": gf_n,
            "chains_per_group": st["chains_per_group"],
            "slip_variables": nvar,
            "distinct_rows_per_patch": {"mean": st["mean_rows"], "max": st["max_rows"],
                                        "of": spec_leg.D * spec_leg.S},
        }
        if ml_runs:
            roof["note"] = ("rows of a cell read from LDS once per run of chains sharing it (4.2 chains per cell on the config-3 population "
                            "with the hypocentre chain order), accumulators through the VGPR index register with ONE scalar instruction per "
                            "chain (s_add_u32 m0, d, d: accumulator and new-cell bit), descriptors by scalar loads, weights-only record "
                            "ring a step ahead; a patch that touches more rows than a 104-slot LDS buffer holds is staged in row passes "
                            "along the duration axis.  Timing-only builds on one box, config 3 (profiles/r5_variants.md, r4_variants.md): 13.4 ms as "
                            "shipped; the consumer wavefronts ALONE (no loaders) 11.7, without their step barrier 10.8, also without row "
                            "reads / scalar loads / records 9.4, without the FMAs 7.8 (the FMAs alone: 7.8 ms at 2 GHz); the loaders' 38 GB "
                            "alone 6.3 (6.1 TB/s).  VALU 54 % busy, scalar unit 45 %, SQ_WAIT_ANY 58 % of the wavefront cycles "
                            "(profiles/r5_runs_sq_counters.json): a wavefront walks a serial path per step and four wavefronts per SIMD "
                            "(74 accumulator registers each) do not fill each other's waits; rows requested a cell ahead were built and "
                            "measured (1.7 % slower, profiles/r5_rows_a_cell_ahead.patch)")
        return roof

    def attach_traffic(roof_d, summary_path):
        """HBM bytes per launch of the leg's kernel from the committed PMC summary of the same command
        (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streams, + WRITE_SIZE;
        separate rocprofv3 --pmc passes run by the builder, tools/run_r3_profile.sh) -- only when the
        kernel that ran here is the one the summary is of"""
        if not os.path.exists(summary_path):
            return
        pmc = json.load(open(summary_path))
        kern = pmc.get("kernel", "").replace("beatamd::", "").split("(")[0].replace(" ", "")
        mine = roof_d["kernel"].replace(" ", "")
        # (the runs kernel reports <epilogue, loader hint>; its symbol is k_gfstack_runs<hint, variant>)
        same = kern == mine or (kern.startswith("k_gfstack_runs<") and mine.startswith("k_gfstack_runs<"))
        if "hbm_read_bytes_per_launch_corrected" not in pmc or not same:
            return
        tr = pmc["hbm_read_bytes_per_launch_corrected"] + pmc.get("hbm_write_bytes_per_launch", 0.0)
        roof_d["traffic"] = tr
        roof_d["traffic_source"] = os.path.relpath(summary_path, ROOT)
        roof_d["traffic_measured_in"] = "rocprofv3 --pmc passes of the same command (tools/run_gpu.sh profile6; not this run)"
        roof_d["hbm_counter_frac"] = tr / (roof_d["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof_d["achieved"] = tr / (roof_d["avg_launch_ms"] * 1e-3) / 1e9
        roof_d["frac"] = roof_d["hbm_counter_frac"]
        roof_d["frac_basis"] = "PMC counter bytes (FETCH_SIZE x 2 + WRITE_SIZE) / HIP-event launch time"

    out = None
    if rank == 0:
        roof = stack_roofline(spec, main_leg, B)
        avg_ms = roof["avg_launch_ms"]
        out = {
            "metric": "SMC chain-steps/s (FFI seismic gfstacking 400 patches x %d targets x %d samples)"
                      % (spec.T, spec.N),
            "value": world * B * K / dt_max,
            "unit": "chain-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt_max / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: FFI seismic gfstacking, 400 patches x %d targets x "
                            "%d samples, library (%d,%d,%d,%d,%d) f64 = %.1f GB in HBM, %s, "
                            "covariance %s%s; population %s"
                            % (spec.T, spec.N, spec.T, spec.P, spec.D, spec.S, spec.N,
                               spec.lib_bytes / 1e9, spec.interpolation, spec.covariance,
                               " (library pre-whitened)" if args.prewhiten else "",
                               "SURVEY 8(d): nucleation U over the 20x20 fault, time 0"
                               if args.prior == "survey" else "round-1 narrow prior (hypocentre in the "
                               "central 7.5 km, time U(0,0.5))"),
                "chains_per_gpu": B,
                "global_chains": world * B,
                "parallelism": "chains sharded over %d GPU(s), library replicated" % world,
                "proposal_step_scale": args.step_scale,
                "env_knobs": {k: os.environ[k] for k in env_knobs},
            },
            "roofline": roof,
            "kernel_ms_per_step": {k: (v[0] / main_leg["steps_timed"]) for k, v in main_leg["times"].items() if v[1]},
            "repeats": R, "region_s": region_s,
            "in_box_fraction": main_leg["in_box"],
            "accept_rate_last_step": main_leg["accept_last"],
            "stage_transition_ms": stage_ms,
            "stage_transition": {"ranks_in_all_gather": world if use_dist else 1, "gathered_chains": int(smc.Q_all.shape[0]),
                                 "population_checksum": {"sum_Q": pop_sum[0], "sum_like": pop_sum[1], "next_beta": pop_sum[2]},
                                 "backend": (backend if use_dist else None)},
            "setup_s": t_build,
        }
        q_ms, q_n = main_leg["times"]["quadform"]
        if q_n:
            # dense-W misfit: chain-batched W.R on the FP64 matrix cores; W upper triangular -> half
            qflops = 2.0 * spec.T * spec.N * spec.N / 2.0 * B
            qa = qflops / (q_ms / q_n * 1e-3) / 1e12
            out["roofline_quadform"] = {
                "bound": "fp64_mfma", "kernel": "k_quadform<128>", "achieved": qa,
                "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": qa / FP64_VALU_PEAK_TFLOPS,
                "avg_launch_ms": q_ms / q_n, "launches": q_n, "flops_per_launch": qflops,
                "mfma_loop_ceiling_TFLOPs": 49.4,
                "note": "v_mfma_f64_16x16x4_f64; a register-resident MFMA loop reaches 49.4 TF on this "
                        "part (tools/micro/mfma64.hip), nominal 78.6"}
        # HBM traffic of the dominant kernel from the PMC passes of the same command
        # (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streams, +WRITE_SIZE)
        default_cfg = (B == 512 and spec.interpolation == "nearest_neighbor" and spec.covariance == "scalar"
                       and spec.T == 64 and spec.N == 4096 and args.gf_order is None and args.prior == "survey"
                       and not env_knobs)
        if default_cfg:
            attach_traffic(roof, args.pmc_summary or os.path.join(args.profiles_dir, PROFILE_ROUND + "_bench_c512_nn_gfstack_ws_summary.json"))
        roof["note"] = ("`bound` names the largest of the fractions; the float-storage leg (half the bytes, same gather: "
                        "same time; half the LDS gather instructions: -9 %) shows the kernel is limited by its LDS-gather "
                        "and row-request instruction counts under a power envelope -- GRBM_GUI_ACTIVE gives 1.72 GHz sustained on "
                        "these values, 1.90 GHz on float-rounded ones for the same cycle count "
                        "(profiles/r3_clock_ws_kernels.json) -- not by HBM bandwidth itself (DESIGN 3.1b)")

    # ---- reuse-free streaming leg: k_gfstack in (chain, target, tile) order, every chain's rows
    # streamed from HBM -- the roofline of SURVEY 8(d)'s algorithmic bytes, driver-observed
    if world == 1 and not args.no_streaming_leg and spec.covariance == "scalar":
        saved = {k: os.environ.get(k) for k in ("BEATAMD_GF_KERNEL", "BEATAMD_GF_ORDER")}
        os.environ["BEATAMD_GF_KERNEL"], os.environ["BEATAMD_GF_ORDER"] = "0", "0"
        ctx.reload_knobs()      # (the context reads its knobs once)
        Bs = min(B, 128)
        leg = run_leg(spec, f, Bs, 4, 1, seed_offset=1000)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        ctx.reload_knobs()
        ms, n = leg["times"]["gfstack"]
        alg_s = algorithmic_bytes_per_chain_step(spec) * Bs
        ach = alg_s / (ms / max(n, 1) * 1e-3) / 1e9
        out["roofline_streaming"] = {
            "bound": "hbm", "kernel": leg["kernel"], "chains": Bs, "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_s,
            "avg_launch_ms": ms / max(n, 1), "launches": n,
            "chain_steps_per_s": Bs * 4 / leg["dt"],
            "traffic": None,
            "note": "no cross-chain row reuse: algorithmic bytes = HBM bytes (block order chain-major)"}
        if Bs == 128 and spec.T == 64 and spec.N == 4096:
            attach_traffic(out["roofline_streaming"], os.path.join(args.profiles_dir, PROFILE_ROUND + "_bench_c512_nn_gfstack0_summary.json"))
    # ---- the same population in larger batches: chain groups of one (target, tile) share an XCD,
    # rows common to several groups come from its L2 (labelled leg; `value` stays the 512-chain batch)
    if world == 1 and not args.no_batch_leg and spec.covariance == "scalar" and B < 2048:
        leg = run_leg(spec, f, 2048, max(K // 2, 3), 1, seed_offset=1000)
        ms, n = leg["times"]["gfstack"]
        out["batch_2048_leg"] = {"chains": 2048, "chain_steps_per_s": 2048 * max(K // 2, 3) / leg["dt"],
                                 "gfstack_avg_launch_ms": ms / max(n, 1), "kernel": leg["kernel"],
                                 "distinct_rows_per_patch_mean": leg["stats"]["mean_rows"]}
    # ---- round 1's narrow prior, labelled, for continuity with BENCH_r01
    if world == 1 and not args.no_narrow_leg and args.prior == "survey" and spec.covariance == "scalar":
        import copy
        a2 = copy.copy(args)
        a2.prior = "narrow"
        spec_n = make_spec(a2)
        from beat_amd.synthetic import _layout_and_bounds
        _, lo_n, up_n = _layout_and_bounds(spec_n)
        host_of[spec_n] = dict(lower=lo_n, upper=up_n)
        leg = run_leg(spec_n, f, B, max(K // 2, 3), 1, seed_offset=1000)
        ms, n = leg["times"]["gfstack"]
        out["narrow_prior_leg"] = {
            "population": "round 1: hypocentre within the central 7.5 km of the fault, time U(0,0.5)",
            "chain_steps_per_s": B * max(K // 2, 3) / leg["dt"], "gfstack_avg_launch_ms": ms / max(n, 1),
            "distinct_rows_per_patch_mean": leg["stats"]["mean_rows"], "kernel": leg["kernel"],
            "hbm_required_bytes_per_launch": leg["stats"]["row_bytes"]}
    # ---- labelled legs of the other configurations (never `value`): same library in HBM, same population
    legs = set(x for x in args.variant_legs.split(",") if x)
    if world == 1 and not args.no_variant_legs and legs and spec.interpolation == "nearest_neighbor" \
            and spec.covariance == "scalar" and not args.prewhiten:
        import copy

        from beat_amd.models.problem import FFIProblem, SeismicWavemap
        from beat_amd.synthetic import exponential_data_covariance
        wm0 = prob.wavemaps[0]
        Kl = max(K // 2, 3)
        N, T = spec.N, spec.T
        f_ml = f_tp = f_pw = f_pm = f_mt = upd = leg = leg2 = leg64 = Lq = L_unrounded = L_rounded = None
        Wd = slog_d = wm_pw = wm_ml = gstep = gQ = gL = gf_ = s_pt = ls_pt = man = st = pop_s = lp_s = None

        def variant(interp="nearest_neighbor", weights=None, slog=None, prewhiten=False):
            """another compiled model over the SAME device library (no second copy unless pre-whitened)"""
            wm = SeismicWavemap(wm0.gfs, wm0.data, wm0.weights if weights is None else weights,
                                wm0.slog_pdet if slog is None else slog, wm0.hypers, wm0.time_shifts, interp)
            pv = FFIProblem(prob.layout, prob.n_patch_dip, prob.n_patch_strike, prob.patch_sizes,
                            prob.slip_varnames, [wm], None, None, prob.lower, prob.upper)
            return pv.compile(ctx, prewhiten=prewhiten)

        def dense_weights():
            """SURVEY 8(d) covariance (ii): sigma_t^2 exp(-|i-j| dt/T0), dt 0.5, T0 2 -> (W [T,N,N], slog [T])"""
            rng_w = np.random.default_rng(spec.seed + 1)
            base = exponential_data_covariance(N, 0.5, 2.0)
            Wb = np.linalg.cholesky(np.linalg.inv(base)).T
            ldb = 2.0 * np.log(np.diag(np.linalg.cholesky(base))).sum()
            scal = (spec.sigma * (1.0 + 0.1 * rng_w.random(T))) ** 2
            Wd = np.empty((T, N, N))
            for t in range(T):
                np.divide(Wb, np.sqrt(scal[t]), out=Wd[t])
            return Wd, np.array([ldb + N * np.log(x) for x in scal])

        def spec_with(**kw):
            a2 = copy.copy(args)
            for k_, v_ in kw.items():
                setattr(a2, k_, v_)
            sp = make_spec(a2)
            host_of[sp] = host
            return sp

        if "multilinear" in legs:
            # the reference's default interpolation (beat/config.py:571-575)
            spec_ml = spec_with(interp="multilinear")
            f_ml = variant("multilinear")
            leg = run_leg(spec_ml, f_ml, B, Kl, 2, seed_offset=1000)
            roof_ml = stack_roofline(spec_ml, leg, B)
            if B == 512 and T == 64 and N == 4096 and args.prior == "survey" and not env_knobs:
                attach_traffic(roof_ml, os.path.join(args.profiles_dir, PROFILE_ROUND + "_bench_c512_ml_gfstack_runs_summary.json"))
            out["multilinear_leg"] = {
                "interpolation": "multilinear (4 rows per patch and chain; beat/ffi/base.py:663-704)",
                "chains": B, "steps": Kl, "chain_steps_per_s": B * Kl / leg["dt"],
                "ms_per_step": leg["dt"] / Kl * 1e3, "roofline": roof_ml,
                "kernel_ms_per_step": {k: (v[0] / Kl) for k, v in leg["times"].items() if v[1]}}
            if B < 2048:
                leg2 = run_leg(spec_ml, f_ml, 2048, 3, 1, seed_offset=1000)
                ms2, n2 = leg2["times"]["gfstack"]
                out["multilinear_leg"]["batch_2048"] = {
                    "chains": 2048, "chain_steps_per_s": 2048 * 3 / leg2["dt"], "kernel": leg2["kernel"],
                    "gfstack_avg_launch_ms": ms2 / max(n2, 1), "gfstack_ms_per_512_chains": ms2 / max(n2, 1) / 4.0}
        if "smc" in legs:
            # the sampler a user runs, end to end (beat/sampler/smc.py:333-546): smc_sample on this problem -- initial
            # stage + 3 tempering stages of 50 Metropolis steps each through the one-call step
            # (beatamd_ffi_mstep_batch: Philox proposals from the population factor, forward model, accept), stage
            # transitions on the device, with and without the stage directories (NumpyChain one-draw traces + state)
            import shutil
            import tempfile

            from beat_amd.sampler import smc_sample
            n_smc = 50
            smc_out = {}
            smc_warm = set()
            runs_ = [(B, False, f, ""), (B, True, f, ""), (2048, False, f, ""), (4096, False, f, ""), (4096, True, f, "")]
            if "multilinear" in legs:
                runs_.append((B, False, f_ml, "_multilinear"))      # the reference's default interpolation, end to end
            for nch, with_files, f_smc, tag_ in runs_:
                if nch > B and B >= 2048:
                    continue
                if (nch, id(f_smc)) not in smc_warm:
                    # untimed: a chain count's first smc_sample of a process carries the stacking kernel's one-off
                    # group-size measurement and the scratch allocations (0.3 s at 2048 chains, 2 s at 4096: tools/time_smc.py)
                    smc_warm.add((nch, id(f_smc)))
                    smc_sample(4, SMC(f_smc, lo, up, n_chains=nch, device=dev, random_seed=12, tune_interval=25), max_stages=1,
                               homepath=None, final_stage=False)
                st = SMC(f_smc, lo, up, n_chains=nch, device=dev, random_seed=11, tune_interval=25)
                home = tempfile.mkdtemp(prefix="beatamd_smc_") if with_files else None
                ctx.enable_timing(True)
                ctx.reset_timing()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pop_s, lp_s, betas_s = smc_sample(n_smc, st, max_stages=3, homepath=home, final_stage=False,
                                                  layout=lay if with_files else None,
                                                  out_names=prob.out_names if with_files else None)
                torch.cuda.synchronize()
                dt_s = time.perf_counter() - t0
                if home:
                    shutil.rmtree(home, ignore_errors=True)
                tmg = dict(st.timings)
                nsteps_s = tmg.pop("steps")
                # HIP-event sums of the context's timers over the whole call, per Metropolis step (the first call of a
                # chain count measures the stacking kernel's group size: `group_size_measurement`, part of `gfstack`)
                ksplit = {k: ctx.kernel_time(k)[0] / nsteps_s for k in ("sweep", "tables", "grouptables", "gfstack", "quadform",
                                                                        "finish", "astep", "proposal", "stage") if ctx.kernel_time(k)[1]}
                ctx.enable_timing(False)
                smc_out["%d_chains%s%s" % (nch, "_with_stage_files" if with_files else "", tag_)] = {
                    "chains": nch, "stages": len(betas_s) - 1, "steps_per_stage": n_smc, "metropolis_steps": nsteps_s,
                    "wall_s": dt_s, "chain_steps_per_s_whole_call": nch * nsteps_s / dt_s,
                    "chain_steps_per_s_sampling_only": nch * nsteps_s / tmg["sample_s"],
                    "split_s": tmg, "kernel_ms_per_step": ksplit, "sampling_ms_per_step": tmg["sample_s"] / nsteps_s * 1e3,
                    "group_size_measurement": ctx.gf_tune_log() if hasattr(ctx, "gf_tune_log") else None,
                    "betas": [float(b_) for b_ in betas_s],
                    "acceptance_per_stage": [float(a_) for a_ in st.stage_acceptance], "finite": bool(np.isfinite(lp_s).all())}
                del st
            smc_out["note"] = ("whole call incl. the initial evaluation of the prior population, transitions, all-gathers and "
                               "(where stated) stage files; `value` above is the same step with torch-drawn proposal rows "
                               "handed in (beatamd_ffi_astep_batch) on the prior population -- later stages concentrate the "
                               "population (fewer distinct rows per patch), which is why sampling-only can exceed `value`")
            out["smc_leg"] = smc_out
        if f_ml is not None:
            f_ml.release()
        f_ml = None
        f_tp = None
        def set_band(on):
            """banded evaluation of upper-triangular whitening operators (beatamd_weights_band) on / off: the dense-W legs
            below time the FP64-MFMA kernel, i.e. with the band OFF; `banded` entries say what the same model does by default"""
            if on:
                os.environ.pop("BEATAMD_QF_BAND", None)
            else:
                os.environ["BEATAMD_QF_BAND"] = "0"
            ctx.reload_knobs()

        if legs & {"toeplitz", "pt", "prewhitened"}:
            Wd, slog_d = dense_weights()
            spec_tp = spec_with(covariance="toeplitz")
            set_band(False)
        if "toeplitz" in legs:
            # chain-batched W.R on the FP64 matrix cores
            f_tp = variant(weights=Wd, slog=slog_d)
            leg = run_leg(spec_tp, f_tp, B, Kl, 2, seed_offset=1000)
            q_ms, q_n = leg["times"]["quadform"]
            qflops = 2.0 * T * N * N / 2.0 * B
            qa = qflops / (q_ms / max(q_n, 1) * 1e-3) / 1e12
            out["toeplitz_leg"] = {
                "covariance": "Toeplitz sigma^2 exp(-|i-j| dt/T0), dt 0.5, T0 2: dense upper-triangular W (8.6 GB) through the dense FP64-MFMA kernel (BEATAMD_QF_BAND=0; the library's default for this operator is the banded evaluation: `banded`)",
                "chains": B, "steps": Kl, "chain_steps_per_s": B * Kl / leg["dt"],
                "ms_per_step": leg["dt"] / Kl * 1e3,
                "kernel_ms_per_step": {k: (v[0] / Kl) for k, v in leg["times"].items() if v[1]},
                "roofline_quadform": {
                    "bound": "fp64_mfma", "kernel": "k_quadform<128>", "achieved": qa,
                    "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": qa / FP64_VALU_PEAK_TFLOPS,
                    "avg_launch_ms": q_ms / max(q_n, 1), "launches": q_n, "flops_per_launch": qflops,
                    "mfma_loop_ceiling_TFLOPs": 49.4, "frac_of_mfma_loop_ceiling": qa / 49.4,
                    "traffic": None,
                    "algorithmic_bytes_per_launch": T * N * N * 4.0 + 2.0 * B * T * N * 8.0,   # W's upper half + R in, once
                    "note": "v_mfma_f64_16x16x4_f64; a register-resident MFMA loop reaches 49.4 TF on this part "
                            "(tools/micro/mfma64.hip), nominal 78.6"}}
            # the same model as the library evaluates it BY DEFAULT: this covariance is the reference's "exponential" noise
            # structure (covariance.py:24-51), a Markov kernel whose whitening operator is bidiagonal up to rounding residue
            set_band(True)
            band = ctx.weights_band(f_tp.problem.wavemaps[0]._wset)
            legb = run_leg(spec_tp, f_tp, B, Kl, 2, seed_offset=1000)
            like_b = f_tp.batch(leg["Q"])[:, -1].clone()
            set_band(False)
            like_d = f_tp.batch(leg["Q"])[:, -1]
            band_dev = float(((like_b - like_d).abs() / like_d.abs()).max().item())
            qb_ms, qb_n = legb["times"]["quadform"]
            out["toeplitz_leg"]["banded"] = {
                "what": "the same weight set evaluated on its band (default: beatamd_weights_band; the dense figures of this leg "
                        "are measured with BEATAMD_QF_BAND=0): W = chol(inv(C)).T of C_ij = exp(-|i-j| dt/T0) is bidiagonal, "
                        "entries beyond the band are <= 2^-40 of the largest (rounding residue of inv + cholesky)",
                "half_bandwidth": band, "chains": B, "steps": Kl, "chain_steps_per_s": B * Kl / legb["dt"],
                "ms_per_step": legb["dt"] / Kl * 1e3,
                "kernel_ms_per_step": {k: (v[0] / Kl) for k, v in legb["times"].items() if v[1]},
                "kernel": legb["kernel"],
                "quadform_kernel": "k_quadform_band1 behind the stacking kernel" if qb_n else
                                   "none: the bidiagonal misfit rides in the stacking kernel's epilogue (mode 3, round 6)",
                "quadform_avg_launch_ms": qb_ms / max(qb_n, 1),
                "quadform_bytes_per_launch": 8.0 * B * T * N + 8.0 * T * N * ((band or 1) + 1) * ((B + 7) // 8),
                "max_rel_dev_of_like_vs_dense": band_dev}
            if B == 512 and T == 64 and N == 4096 and not env_knobs:
                qsum = os.path.join(ROOT, "profiles", "r4_bench_c512_toeplitz_quadform128_summary.json")
                if os.path.exists(qsum):
                    pq = json.load(open(qsum))
                    rq = out["toeplitz_leg"]["roofline_quadform"]
                    rq["traffic"] = pq["hbm_read_bytes_per_launch_corrected"] + pq.get("hbm_write_bytes_per_launch", 0.0)
                    rq["traffic_source"] = os.path.relpath(qsum, ROOT)
                    rq["traffic_measured_in"] = "builder rocprofv3 --pmc passes of `bench.py --covariance toeplitz` (not this run)"
            if "default_config" in legs:
                # the configuration a real FFI run uses: the reference's default interpolation (multilinear,
                # beat/config.py:571-575) TOGETHER with a non-diagonal data covariance (beat/covariance.py:397-427,
                # models/seismic.py:1509-1534): stacking with the residual-store epilogue + dense W on the matrix cores
                f_mt = variant("multilinear", weights=Wd, slog=slog_d)
                spec_mt = spec_with(interp="multilinear", covariance="toeplitz")
                leg = run_leg(spec_mt, f_mt, B, Kl, 2, seed_offset=1000)
                out["default_config_leg"] = {"multilinear_dense_W": {
                    "configuration": "multilinear interpolation + Toeplitz data covariance (dense upper-triangular W, 8.6 GB)",
                    "chains": B, "steps": Kl, "chain_steps_per_s": B * Kl / leg["dt"], "ms_per_step": leg["dt"] / Kl * 1e3,
                    "kernel": leg["kernel"],
                    "kernel_ms_per_step": {k: (v[0] / Kl) for k, v in leg["times"].items() if v[1]}}}
                set_band(True)
                legb = run_leg(spec_mt, f_mt, B, Kl, 2, seed_offset=1000)
                set_band(False)
                out["default_config_leg"]["multilinear_banded_W"] = {
                    "configuration": "the same model with the weight set evaluated on its band (the library's default)",
                    "chains": B, "steps": Kl, "chain_steps_per_s": B * Kl / legb["dt"], "ms_per_step": legb["dt"] / Kl * 1e3,
                    "kernel": legb["kernel"],
                    "kernel_ms_per_step": {k: (v[0] / Kl) for k, v in legb["times"].items() if v[1]}}
                f_mt.release()
                f_mt = None
        if "pt" in legs:
            # parallel tempering: 4 temperatures x 256 replicas = the per-GPU share of BASELINE configs[4];
            # the sampler times its own rounds (set-up of the replicas excluded)
            from beat_amd.sampler import pt_sample
            n_rep, n_temp = 256, 4
            kw = dict(n_chains_posterior=1, n_chains_tempered=n_temp - 1, n_replicas=n_rep, swap_interval=(3, 5),
                      beta_tune_interval=4, proposal_cov=np.diag(((up - lo) * args.step_scale) ** 2), device=dev,
                      random_seed=5)
            # ("toeplitz": the dense kernel as in the rounds before, BEATAMD_QF_BAND=0; "toeplitz_banded": the library's default
            # for this operator)
            for cov_name, f_pt in (("scalar", f), ("toeplitz", f_tp), ("toeplitz_banded", f_tp)):
                if f_pt is None:
                    continue
                set_band(cov_name == "toeplitz_banded")
                pt_sample(f_pt, lo, up, n_samples=n_rep, **kw)   # warm-up: allocations, the measured group size
                ctx.enable_timing(True)
                ctx.reset_timing()
                s_pt, ls_pt, man = pt_sample(f_pt, lo, up, n_samples=8 * n_rep, **kw)
                g_ms, n_launch = ctx.kernel_time("gfstack")
                ctx.enable_timing(False)
                out.setdefault("pt_leg", {})[cov_name] = {
                    "replicas": "%d temperatures x %d replicas = %d chains on one GPU (the per-GPU share of BASELINE "
                                "configs[4]), exchange round every 3-5 steps" % (n_temp, n_rep, n_temp * n_rep),
                    "steps": man.loop_steps, "exchange_rounds": man._round,
                    "chain_steps_per_s": n_temp * n_rep * man.loop_steps / man.loop_seconds,
                    "ms_per_step_incl_exchange": man.loop_seconds / man.loop_steps * 1e3,
                    "gfstack_avg_launch_ms": g_ms / max(n_launch, 1), "finite": bool(np.isfinite(ls_pt).all())}
            set_band(False)
        if "stage_update" in legs and f_tp is not None:
            # what update_covariances costs at a stage boundary (smc.py:492-503): covariance re-estimation at the
            # MAP point + factorisation + new weights installed, then the end points evaluated again
            from beat_amd.covariance import NoiseCovarianceUpdate
            upd = NoiseCovarianceUpdate(f_tp)
            q_map = leg_q = None
            Lq = f_tp.batch(main_leg["Q"])
            q_map = main_leg["Q"][int(torch.argmax(Lq[:, -1]))].cpu().numpy()
            for rep in range(3):      # (the first two calls allocate: 8.6 GB buffers through the caching allocator)
                upd.update_weights(q_map)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                f_tp.batch(main_leg["Q"])
                torch.cuda.synchronize()
                t_re = time.perf_counter() - t0
            out["stage_update_leg"] = {"dense_W": {
                "what": "64 noise covariances of 4096^2 re-estimated at the MAP point (synthetics, running rms, "
                        "autocovariance, scaled Toeplitz), factorised (= the PSD test), weights installed; then the %d end "
                        "points evaluated again" % B,
                "stage_update_s": upd.last_ms * 1e-3 + t_re, "update_weights_s": upd.last_ms * 1e-3,
                "reevaluation_s": t_re, "repaired_on_host": upd.n_repaired}}
        upd = Lq = None
        if f_tp is not None:
            f_tp.release()
        f_tp = None
        torch.cuda.empty_cache()
        if "prewhitened" in legs:
            # W.G and W.d computed once, no dense W.r per step (needs a second copy of the library)
            try:
                t0 = time.perf_counter()
                f_pw = variant(weights=Wd, slog=slog_d, prewhiten=True)
                torch.cuda.synchronize()
                t_pw = time.perf_counter() - t0
                leg = run_leg(spec_tp, f_pw, B, Kl, 2, seed_offset=1000)
                out["prewhitened_leg"] = {
                    "covariance": "the Toeplitz covariance folded into a whitened COPY of the library (W.G, W.d once)",
                    "chains": B, "steps": Kl, "chain_steps_per_s": B * Kl / leg["dt"],
                    "ms_per_step": leg["dt"] / Kl * 1e3, "whitening_s": t_pw, "kernel": leg["kernel"]}
                if "default_config" in legs:
                    # multilinear on the SAME whitened copy (no dense W.r per step)
                    wm_pw = f_pw.problem.wavemaps[0]
                    wm_ml = SeismicWavemap(wm_pw.gfs, wm_pw.data, wm_pw.weights, wm_pw.slog_pdet, wm_pw.hypers,
                                           wm_pw.time_shifts, "multilinear")
                    f_pm = FFIProblem(prob.layout, prob.n_patch_dip, prob.n_patch_strike, prob.patch_sizes,
                                      prob.slip_varnames, [wm_ml], None, None, prob.lower, prob.upper).compile(ctx)
                    leg = run_leg(spec_with(interp="multilinear", covariance="toeplitz"), f_pm, B, Kl, 2, seed_offset=1000)
                    out.setdefault("default_config_leg", {})["multilinear_prewhitened"] = {
                        "configuration": "multilinear interpolation on the pre-whitened library (Toeplitz covariance folded in)",
                        "chains": B, "steps": Kl, "chain_steps_per_s": B * Kl / leg["dt"],
                        "ms_per_step": leg["dt"] / Kl * 1e3, "kernel": leg["kernel"]}
                    f_pm.release()
                    f_pm = None
                if "stage_update" in legs:
                    # on the pre-whitened path an update re-whitens all 62.9 GB of rows in place (M = W_new inv(W_old))
                    from beat_amd.covariance import NoiseCovarianceUpdate
                    upd = NoiseCovarianceUpdate(f_pw)
                    Lq = f_pw.batch(main_leg["Q"])
                    q_map = main_leg["Q"][int(torch.argmax(Lq[:, -1]))].cpu().numpy()
                    for rep in range(3):
                        ctx.enable_timing(True)
                        ctx.reset_timing()
                        upd.update_weights(q_map)
                        torch.cuda.synchronize()
                        w_ms, w_n = ctx.kernel_time("whiten")
                        ctx.enable_timing(False)
                        t0 = time.perf_counter()
                        f_pw.batch(main_leg["Q"])
                        torch.cuda.synchronize()
                        t_re = time.perf_counter() - t0
                    wflops = float(spec.T) * spec.P * spec.D * spec.S * N * N
                    out.setdefault("stage_update_leg", {})["prewhitened"] = {
                        "what": "the same update on the pre-whitened model: residuals un-whitened (one back substitution per dataset), covariances, factorisation, "
                                "M = W_new inv(W_old), all %.1f GB of library rows re-whitened in place (beatamd_whiten_rows_batch), "
                                "end points evaluated again" % (spec.lib_bytes / 1e9),
                        "stage_update_s": upd.last_ms * 1e-3 + t_re, "update_weights_s": upd.last_ms * 1e-3,
                        "rewhitening_s": w_ms * 1e-3, "rewhitening_TFLOPs": wflops / (w_ms * 1e-3) / 1e12 if w_ms else None,
                        "reevaluation_s": t_re}
                upd = Lq = None
                f_pw.release()
                f_pw = None
            except (RuntimeError, MemoryError) as exc:   # not enough HBM for the copy beside other allocations
                out["prewhitened_leg"] = {"skipped": str(exc)[:200]}
            torch.cuda.empty_cache()
        set_band(True)       # (the dense-W legs above ran with the band off)
        if "fp32" in legs:
            # float-storage library (SURVEY 8(f) row 2 "optional fp32 layout"): float copy of the library, the
            # float64 storage rounded to the same values (LAST leg on the shared library for that reason); rows
            # move as 256-byte segments, operands are widened before the f64 FMA, accumulation stays f64
            L_unrounded = f.batch(main_leg["Q"]).clone()
            f.round_libraries_to_f32()      # explicit, irreversible: from here on the shared library holds rounded values
            torch.cuda.synchronize()
            L_rounded = f.batch(main_leg["Q"])
            rel_err = ((L_rounded - L_unrounded).abs() / L_unrounded.abs()).max(0).values
            leg = run_leg(spec, f, B, Kl, 2, seed_offset=1000)
            ms32, n32 = leg["times"]["gfstack"]
            # the float64 kernel on the SAME (now float-representable) values: separates what the data does
            # (zero low mantissa bits: less switching power, higher sustained clocks) from what the kernel does
            f.set_f32(False)
            leg64 = run_leg(spec, f, B, Kl, 2, seed_offset=1000)
            ms64, n64 = leg64["times"]["gfstack"]
            need32 = main_leg["stats"]["row_bytes"] / 2.0
            out["fp32_storage_leg"] = {
                "storage": "library rows as float32 (31.5 GB copy; values rounded by <= 6e-8 relative), f64 accumulation, "
                           "weights and likelihood; NOT the precision of `value`",
                "chains": B, "steps": Kl, "chain_steps_per_s": B * Kl / leg["dt"], "ms_per_step": leg["dt"] / Kl * 1e3,
                "kernel": leg["kernel"], "gfstack_avg_launch_ms": ms32 / max(n32, 1),
                "f64_kernel_on_the_rounded_library": {"kernel": leg64["kernel"],
                                                      "gfstack_avg_launch_ms": ms64 / max(n64, 1),
                                                      "chain_steps_per_s": B * Kl / leg64["dt"]},
                "hbm_required_bytes_per_launch": need32,
                "hbm_frac_required_bytes": need32 / (ms32 / max(n32, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "max_rel_err_vs_unrounded_library": {"like": float(rel_err[-1]), "dataset_logpts": float(rel_err[:-1].max()),
                                                     "chains": int(L_rounded.shape[0]),
                                                     "north_star_tolerance": 1e-6},
                "note": "float pairs gathered by ds_read_b64: half the bytes and half the LDS gather instructions of the "
                        "f64 kernel for 2-4 % -- the rest of the difference to `value` is the DATA: the f64 kernel itself "
                        "runs 4-7 % faster on float-representable values (zero low mantissa bits, less switching "
                        "power, higher sustained clocks).  The kernel is limited by its instruction streams and the "
                        "power envelope, not by HBM bytes (DESIGN 3.1b)"}
        if "geometry" in legs:
            # geometry mode, BASELINE configs[1]: rectangular source, 2 SAR scenes (214 + 205 points, full
            # covariances), 1024 SMC chains; synthetic observations; parity with BEAT unpinned (pyrocko absent)
            from beat_amd.synthetic import build_geometry_problem
            gprob, glay, glower, gupper = build_geometry_problem()
            glo, gup = glay.bounds(glower, gupper)
            gf_ = gprob.compile(ctx)
            gleg = {}
            for use_graph in (False, True):
                gstep = SMC(gf_, glo, gup, n_chains=1024, tune_interval=10, device=dev, random_seed=2,
                            use_graph=use_graph)
                gQ = gstep.initialize_population()
                gL = gstep.stepper.evaluate(gQ)
                gstep.select_end_points(gQ, gL)
                best = None
                for stage in range(3):
                    gstep.transition()
                    gstep.stage += 1
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    gQ, gL = gstep.sample_stage(200)
                    torch.cuda.synchronize()
                    dt_g = time.perf_counter() - t0
                    gstep.select_end_points(gQ, gL)
                    if stage > 0:
                        best = dt_g if best is None else min(best, dt_g)
                gleg["graph" if use_graph else "eager"] = best / 200 * 1e6
            out["geometry_leg"] = {
                "workload": "BASELINE configs[1] shape: rectangular source (Okada 1985), 2 SAR scenes 214 + 205 points "
                            "with full covariances, 1024 SMC chains, synthetic observations",
                "parity": "vs BEAT unpinned (pyrocko's layered GF engine is not in the reference tree); pinned to "
                          "Okada's published check values",
                "launches_per_step": 4,
                "us_per_step_eager": gleg["eager"], "us_per_step_hip_graph": gleg["graph"],
                "chain_steps_per_s": 1024 / (min(gleg.values()) * 1e-6)}
        if "sharded" in legs:
            # libraries sharded by TARGET (SURVEY 8(e) fallback for libraries beyond one GPU): tools/time_sharded.py in
            # processes of their own -- 1 rank replicated (fused step), then 2 and 8 ranks holding the rows of their targets,
            # ALL ON THIS ONE GPU (gloo, the all-gather staged through host memory): the step in pieces + one all-gather
            import subprocess
            sh = {}
            tool = os.path.join(ROOT, "tools", "time_sharded.py")
            env_s = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
            for nr in (1, 2, 8):
                cmd = [sys.executable, tool] if nr == 1 else \
                    [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nr), "--master-addr",
                     "127.0.0.1", "--master-port", str(29650 + nr), tool]
                try:
                    r_ = subprocess.run(cmd + ["--targets", "16", "--samples", "2048", "--chains", str(B), "--steps", "10"],
                                        env=env_s, cwd=ROOT, capture_output=True, text=True, timeout=600)
                    ln = [x for x in r_.stdout.splitlines() if x.startswith("{")]
                    sh["%d_rank%s" % (nr, "" if nr == 1 else "s")] = json.loads(ln[-1]) if ln else {"failed": r_.stderr[-300:]}
                except (subprocess.TimeoutExpired, OSError) as exc:
                    sh["%d_ranks" % nr] = {"failed": str(exc)[:200]}
            rep_ = sh.get("1_rank", {})
            for k_, v_ in sh.items():
                if k_ != "1_rank" and "ms_per_step" in v_ and "ms_per_step" in rep_:
                    v_["chain_steps_per_s"] = B / (v_["ms_per_step"] * 1e-3)
                    v_["overhead_vs_replicated_without_allgather"] = v_["ms_per_step_without_allgather"] / rep_["ms_per_step"] - 1.0
                    v_["bitwise_equal_to_replicated"] = v_["state_checksum"] == rep_["state_checksum"]
            if "ms_per_step" in rep_:
                rep_["chain_steps_per_s"] = B / (rep_["ms_per_step"] * 1e-3)
            sh["workload"] = ("config 3 with 16 targets x 2048 samples (7.9 GB), %d chains, 10 steps; all ranks on one GPU, gloo" % B)
            out["sharded_leg"] = sh
        if legs & {"config4", "realistic_grid"}:
            # legs with libraries of their own: the variant models of the main library (dense weight sets of 8.6 GB each,
            # the whitened library copy, update buffers) are not needed any more
            import gc
            for f_old in (f_ml, f_tp, f_pw):
                if f_old is not None:
                    f_old.release()
            f_ml = f_tp = f_pw = f_pm = f_mt = upd = leg = leg2 = leg64 = Lq = L_unrounded = L_rounded = None   # noqa: F841
            Wd = slog_d = wm_pw = wm_ml = gstep = gQ = gL = gf_ = s_pt = ls_pt = man = st = pop_s = lp_s = None     # noqa: F841
            gc.collect()
            torch.cuda.empty_cache()

        def own_library_leg(spec_l, chains_list, interps, n_steps, fix_problem=None, traffic_tag=None):
            """build a problem with its own device library, time `chains_list` x `interps` on it, free it"""
            res = {}
            t0 = time.perf_counter()
            prob_l, host_l = build_problem(spec_l, device_library=True, ctx=ctx)
            if fix_problem is not None:
                fix_problem(prob_l, host_l)
            host_of[spec_l] = host_l
            nvar_l = len(spec_l.slip_varnames)
            for interp in interps:
                import copy as _copy
                sp_i = _copy.copy(spec_l)
                sp_i.interpolation = interp
                host_of[sp_i] = host_l
                for wm in prob_l.wavemaps:
                    wm.interpolation = interp
                f_i = prob_l.compile(ctx)      # (the libraries are uploaded / adopted once: init_optimization keeps lib_id)
                torch.cuda.synchronize()
                for nch in chains_list:
                    # (timed twice, the faster kept: the first pass after a library change carries one-off allocator work --
                    # a 30 ms hiccup is half of a 12-step leg of 3.5 ms steps)
                    leg = min((run_leg(sp_i, f_i, nch, n_steps, 3, seed_offset=1000) for _ in range(2)), key=lambda l_: l_["dt"])
                    roof_l = stack_roofline(sp_i, leg, nch)
                    if traffic_tag and not env_knobs:
                        attach_traffic(roof_l, os.path.join(args.profiles_dir, "%s_%s_c%d_%s_gfstack_%s_summary.json" % (
                            PROFILE_ROUND, traffic_tag, nch, "nn" if interp == "nearest_neighbor" else "ml",
                            "ws" if interp == "nearest_neighbor" else "runs")))
                    plan = ctx.gf_plan() if hasattr(ctx, "gf_plan") else None
                    wband = [ctx.weights_band(wm._wset) for wm in prob_l.wavemaps if getattr(wm, "_wset", None) is not None]
                    res["%s_%d_chains" % ("nn" if interp == "nearest_neighbor" else "multilinear", nch)] = {
                        "chains": nch, "steps": n_steps, "chain_steps_per_s": nch * n_steps / leg["dt"],
                        "ms_per_step": leg["dt"] / n_steps * 1e3, "kernel": leg["kernel"], "plan": plan,
                        "whitening_operator_half_bandwidth": wband,     # (-1: dense kernel / scalar weights)
                        "gfstack_avg_launch_ms": roof_l["avg_launch_ms"],
                        "gfstack_ms_per_512_chains": roof_l["avg_launch_ms"] * 512.0 / nch,
                        "kernel_ms_per_step": {k: (v[0] / n_steps) for k, v in leg["times"].items() if v[1]},
                        "roofline": roof_l}
                    del leg
                f_i.release()
                del f_i
            res["setup_s"] = time.perf_counter() - t0
            res["library"] = "%d slip variable(s) x (%d,%d,%d,%d,%d) f64 = %.1f GB each" % (
                nvar_l, spec_l.T, spec_l.P, spec_l.D, spec_l.S, spec_l.N, spec_l.lib_bytes / 1e9)
            del prob_l, host_l
            gc.collect()
            torch.cuda.empty_cache()
            return res

        if "config4" in legs:
            # BASELINE configs[3] (SURVEY 8(d) "config 4", reference test/test_ffi_gfstacking_multifault.py:16-122):
            # 2 subfaults of 10 x 20 patches (2 km), 35 targets, two slip components, station time shifts, joint with
            # the geodetic composite (the Laquila SAR scenes: 2 x 214 points, full covariances), Toeplitz data
            # covariance (dense W); 512 chains = the per-GPU share of 4096 chains over 8 GPUs
            from beat_amd.heart import whitening
            from beat_amd.models.problem import GeodeticData
            from beat_amd.synthetic import SyntheticSpec
            gold = os.path.join(ROOT, "tests", "golden", "laquila_geodetic.npz")
            out["config4_leg"] = {}
            for N4 in (4096, 120):
                if os.path.exists(gold):
                    gz = np.load(gold)
                    sizes4 = tuple(int(gz["d%d_displacement" % i].size) for i in range(int(gz["n"])))
                else:
                    gz, sizes4 = None, (214, 214)
                sp4 = SyntheticSpec((10, 10), (20, 20), (2.0, 2.0), T=35, N=N4, D=2, S=60, st_dt=0.5,
                                    slip_varnames=("uparr", "uperp"), covariance="toeplitz", station_shifts=True,
                                    geodetic_nobs=sizes4, vel_bounds=(3.0, 4.0), time_bounds=(0.0, 2.0))

                def laquila(prob_l, host_l, gz=gz, sizes4=sizes4):
                    if gz is None:
                        return
                    gd = prob_l.geodetic
                    d4 = np.concatenate([gz["d%d_displacement" % i] for i in range(len(sizes4))])
                    o4 = np.concatenate([gz["d%d_odw" % i] for i in range(len(sizes4))])
                    ws = [whitening(gz["d%d_C" % i]) for i in range(len(sizes4))]
                    prob_l.geodetic = GeodeticData(gd.gfs, d4, o4, sizes4, [w_[0] for w_ in ws], [w_[1] for w_ in ws],
                                                   gd.hypers)
                r4 = own_library_leg(sp4, (512,), ("multilinear", "nearest_neighbor"), max(Kl // 2, 3), laquila,
                                     traffic_tag="config4_N%d" % N4)
                r4["workload"] = ("BASELINE configs[3]: joint seismic + geodetic FFI, 2 subfaults x (10 x 20) patches of 2 km, 35 "
                                  "targets x %d samples, slip components uparr + uperp, station time shifts, Toeplitz data "
                                  "covariance (dense W), %s; 512 chains = the per-GPU share of 4096 over 8 GPUs"
                                  % (N4, "Laquila SAR scenes (2 x 214 points, full covariances)" if gz is not None
                                     else "synthetic geodetic scenes"))
                out["config4_leg"]["N%d" % N4] = r4
        if "realistic_grid" in legs:
            # a library on the (duration x start-time) grid of the reference's own tutorial and defaults: durations 0-4 s every
            # 0.25 s (docs/examples/FFI_kinematic.rst:185-195), start times 0-20 s every 0.5 s (beat/config.py:1906-1909);
            # the prior spans the duration axis (the library is built from the prior bounds, beat/ffi/base.py:1005-1254)
            from beat_amd.synthetic import SyntheticSpec
            spr = SyntheticSpec((20,), (20,), (1.0,), T=64, N=512, D=17, S=41, st_min=0.0, st_dt=0.5, du_min=0.0, du_dt=0.25,
                                nuc_margin=0.0, time_bounds=(0.0, 0.0))
            rr = own_library_leg(spr, (512, 2048), ("nearest_neighbor", "multilinear"), max(Kl // 2, 3), traffic_tag="grid")
            rr["workload"] = ("one 20 x 20 subfault, 64 targets x 512 samples, durations 0-4 s @ 0.25 s (D=17) x start times 0-20 s "
                              "@ 0.5 s (S=41): library (64,400,17,41,512) f64 = %.1f GB; SURVEY 8(d) population with the duration "
                              "prior spanning the library axis (U(0,4) s)" % (spr.lib_bytes / 1e9))
            out["realistic_grid_leg"] = rr
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        # the complete object (every leg with its notes, plans, splits) goes to disk; stdout ends with ONE compact line
        full_path = None
        try:
            with open(args.full_json, "w") as fh:
                json.dump(out, fh, indent=1)
            full_path = args.full_json
            gdir = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(gdir) and os.path.dirname(os.path.abspath(args.full_json)) != gdir:
                with open(os.path.join(gdir, os.path.basename(args.full_json)), "w") as fh:
                    json.dump(out, fh, indent=1)
        except OSError as exc:
            print("bench.py: full result not written (%s)" % exc, file=sys.stderr)
        sys.stderr.flush()
        print(compact_line(out, full_path), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
