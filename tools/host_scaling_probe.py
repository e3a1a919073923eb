#!/usr/bin/env python
"""host_scaling_probe.py -- what the HOST of the GPU box gives N forked single-thread workers (the reference's
iter_parallel_chains model, beat/sampler/base.py:428-595), measured without any of our code in the loop:

  * the cgroup CPU quota of the container (cpu.max) and the CPUs the process may run on;
  * compute scaling: every worker spins on a cache-resident numpy kernel -- aggregate rate vs one worker;
  * memory scaling: every worker streams its OWN private 256 MB array (first-touched by itself: NUMA-local, no sharing,
    no page-table contention between processes) -- aggregate GB/s vs one worker.

bench.py's cpu_baseline leg scales x3.6 from 1 to 256 workers on the round-5 box although every worker is pinned and
reads a NUMA-local copy of the library; this probe says whether the host itself scales.

    python tools/host_scaling_probe.py [seconds per point]
"""
import json
import os
import sys
import time
from multiprocessing import get_context

import numpy as np


def _worker(kind, cpu, budget, barrier, q):
    try:
        os.sched_setaffinity(0, {cpu})
    except OSError:
        pass
    if kind == "compute":
        a = np.random.default_rng(cpu).random(4096)       # 32 KB: L1 / L2 resident
        unit = a.size * 8
    else:
        a = np.ones(32 * 1024 * 1024)                     # 256 MB, first-touched here
        unit = a.nbytes
    barrier.wait()
    t0 = time.perf_counter()
    n = 0
    s = 0.0
    while time.perf_counter() - t0 < budget:
        if kind == "compute":
            for _ in range(200):
                s += float(np.dot(a, a))
            n += 200
        else:
            s += float(a.sum())
            n += 1
    q.put((n * unit, time.perf_counter() - t0, s))


def point(kind, cpus, budget):
    mp = get_context("fork")
    barrier = mp.Barrier(len(cpus))
    q = mp.Queue()
    procs = [mp.Process(target=_worker, args=(kind, c, budget, barrier, q)) for c in cpus]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join()
    return sum(r[0] for r in res) / max(r[1] for r in res) / 1e9      # GB/s of operand bytes


def probe(budget=2.0, counts=None):
    cpus = sorted(os.sched_getaffinity(0))
    out = {"cpus_allowed": len(cpus)}
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            out[os.path.basename(f)] = open(f).read().strip()
        except OSError:
            pass
    try:
        out["loadavg"] = open("/proc/loadavg").read().split()[:3]
    except OSError:
        pass
    counts = counts or sorted({1, min(16, len(cpus)), min(64, len(cpus)), len(cpus)})
    for kind in ("compute", "memory"):
        pts = {}
        for n in counts:
            sel = cpus[::max(1, len(cpus) // n)][:n]          # spread over the sockets
            pts[str(n)] = point(kind, sel, budget)
        out[kind + "_GBs_by_workers"] = pts
        out[kind + "_scaling_all_vs_1"] = pts[str(counts[-1])] / pts[str(counts[0])]
    return out


if __name__ == "__main__":
    print(json.dumps(probe(float(sys.argv[1]) if len(sys.argv) > 1 else 2.0), indent=1))
