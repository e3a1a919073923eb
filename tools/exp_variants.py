"""A/B runner for the stacking kernel: builds the config-3 problem ONCE (62.9 GB library in HBM) and
times the fused astep under a list of variants (environment knobs of csrc/gfshared.hip, chain
counts, prior, interpolation).  One JSON line per variant on stdout / in the output file.

    python tools/exp_variants.py out.jsonl [variant-file.json]

variant = {"name":..., "env": {"BEATAMD_GS_NT": "32"}, "chains": 512, "prior": "survey",
           "interp": "nearest_neighbor", "steps": 6}"""
import json
import os

os.environ.setdefault("BEATAMD_KNOBS_LIVE", "1")   # this tool flips the BEATAMD_G* knobs between launches
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import beat_amd  # noqa: E402
from beat_amd.synthetic import SyntheticSpec, _layout_and_bounds, build_problem, draw_population  # noqa: E402

DEFAULT = [
    {"name": "default_c512"},
    {"name": "nt32_c512", "env": {"BEATAMD_GS_NT": "32"}},
    {"name": "cg1024_c1024", "env": {"BEATAMD_GS_CG": "1024"}, "chains": 1024},
    {"name": "default_c1024", "chains": 1024},
    {"name": "order0_c1024", "env": {"BEATAMD_GS_ORDER": "0"}, "chains": 1024},
    {"name": "nt32_c1024", "env": {"BEATAMD_GS_NT": "32"}, "chains": 1024},
    {"name": "default_c2048", "chains": 2048},
    {"name": "order0_c2048", "env": {"BEATAMD_GS_ORDER": "0"}, "chains": 2048},
    {"name": "cg1024_c2048", "env": {"BEATAMD_GS_CG": "1024"}, "chains": 2048},
    {"name": "default_c256", "chains": 256},
    {"name": "default_c128", "chains": 128},
    {"name": "narrow_c512", "prior": "narrow"},
    {"name": "narrow_nt32_c512", "prior": "narrow", "env": {"BEATAMD_GS_NT": "32"}},
    {"name": "narrow_cg1024_c1024", "prior": "narrow", "env": {"BEATAMD_GS_CG": "1024"}, "chains": 1024},
    {"name": "ml_c512", "interp": "multilinear"},
    {"name": "ml_nt32_c512", "interp": "multilinear", "env": {"BEATAMD_GS_NT": "32"}},
]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "variants.jsonl")
    variants = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else DEFAULT
    dev = torch.device("cuda", 0)
    ctx = beat_amd.get_context(0)
    ctx.use_torch_stream()
    specs, fs = {}, {}

    def spec_for(prior, interp):
        kw = dict(nuc_margin=0.0, time_bounds=(0.0, 0.0)) if prior == "survey" \
            else dict(nuc_margin=6.0, time_bounds=(0.0, 0.5))
        return SyntheticSpec((20,), (20,), (1.0,), T=64, N=4096, D=3, S=25, interpolation=interp, **kw)

    base = spec_for("survey", "nearest_neighbor")
    prob, host = build_problem(base, device_library=True, ctx=ctx)
    fs["nearest_neighbor"] = prob.compile(ctx)
    lay = host["layout"]
    fh = open(out_path, "a")
    for v in variants:
        interp = v.get("interp", "nearest_neighbor")
        if interp not in fs:
            prob.wavemaps[0].interpolation = interp
            fs[interp] = prob.compile(ctx)
        f = fs[interp]
        spec = spec_for(v.get("prior", "survey"), interp)
        _, lo_n, up_n = _layout_and_bounds(spec)
        C, K = int(v.get("chains", 512)), int(v.get("steps", 6))
        saved = {}
        for k, val in v.get("env", {}).items():
            saved[k] = os.environ.get(k)
            os.environ[k] = val
        rec = dict(name=v["name"], chains=C, steps=K, interp=interp, prior=v.get("prior", "survey"),
                   env=v.get("env", {}))
        try:
            Q0 = torch.from_numpy(draw_population(spec, lay, lo_n, up_n, C)).to(dev)
            lo, up = lay.bounds(lo_n, up_n)
            lo_d, up_d = torch.from_numpy(lo).to(dev), torch.from_numpy(up).to(dev)
            L0 = f.batch(Q0)
            gen = torch.Generator(device=dev)
            gen.manual_seed(4242)
            delta = torch.randn((K + 1, C, lay.size), generator=gen, device=dev, dtype=torch.float64) \
                * (5e-4 * (up_d - lo_d))
            log_u = torch.log(torch.rand((K + 1, C), generator=gen, device=dev, dtype=torch.float64))
            sc = torch.ones(C, device=dev, dtype=torch.float64)
            acc = torch.zeros(C, device=dev, dtype=torch.int32)
            f.astep_batch(Q0, L0, delta[0], sc, lo_d, up_d, log_u[0], 2e-6, acc)
            ctx.synchronize()
            ctx.enable_timing(True)
            ctx.reset_timing()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(1, K + 1):
                f.astep_batch(Q0, L0, delta[i], sc, lo_d, up_d, log_u[i], 2e-6, acc)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ms, n = ctx.kernel_time("gfstack")
            gt, _ = ctx.kernel_time("grouptables")
            ctx.enable_timing(False)
            st = ctx.gf_group_stats()
            rec.update(kernel=ctx.last_kernel(), gfstack_ms=ms / max(n, 1), grouptables_ms=gt / max(n, 1),
                       ms_per_step=dt / K * 1e3, chain_steps_per_s=C * K / dt, mean_rows=st["mean_rows"],
                       max_rows=st["max_rows"], row_GB=st["row_bytes"] / 1e9,
                       hbm_TBs=st["row_bytes"] / (ms / max(n, 1) * 1e-3) / 1e12 if n else None,
                       like_sum=float(L0[:, -1].sum().item()))
            del delta, log_u, Q0, L0
        except Exception as e:  # a variant that fails must not take the others with it
            rec["error"] = "%s: %s" % (type(e).__name__, e)
        for k, val in saved.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
        line = json.dumps(rec)
        print(line, flush=True)
        fh.write(line + "\n")
        fh.flush()


if __name__ == "__main__":
    main()
