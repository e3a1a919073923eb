import os, sys, time
os.environ.setdefault("BEATAMD_KNOBS_LIVE", "1")   # this tool flips BEATAMD_SWEEP_V1 between launches
sys.path.insert(0, "/root/repo")
import numpy as np, torch, beat_amd
ctx = beat_amd.get_context(0)
rng = np.random.default_rng(1)
C, nd, ns = 512, 20, 20
slow = torch.from_numpy(1.0 / rng.uniform(2.5, 4.0, (C, nd * ns))).cuda()
hd = torch.from_numpy(rng.integers(0, nd, C).astype(np.int32)).cuda()
hs = torch.from_numpy(rng.integers(0, ns, C).astype(np.int32)).cuda()
for env in ("0", "1", "0", "1"):
    os.environ["BEATAMD_SWEEP_V1"] = env
    out = ctx.fast_sweep_batch(slow, 1.0, hd, hs, nd, ns)
    ctx.synchronize(); ctx.enable_timing(True); ctx.reset_timing()
    for _ in range(20): out = ctx.fast_sweep_batch(slow, 1.0, hd, hs, nd, ns)
    ctx.synchronize()
    ms, n = ctx.kernel_time("sweep"); ctx.enable_timing(False)
    print("SWEEP_V1=%s: %.1f us per launch (%d)" % (env, ms / n * 1e3, n), float(out.sum()))
