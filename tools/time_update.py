"""Stage-update time of the noise-covariance re-estimation (update= of smc_sample; reference
sampler/smc.py:492-503 -> seismic.py:1509-1534 -> covariance.py:307-427) at the size of BASELINE
configs[2]: 64 datasets x 4096 samples, the 62.9 GB library resident in HBM.  Prints the time of
NoiseCovarianceUpdate.update_weights (synthetics at the MAP point, residuals, running-window rms,
autocovariance, scaled Toeplitz, factorisation = PSD test, weights installed) and of re-evaluating
the 512 end points with the new weights.

    python tools/time_update.py [chains=512] [prewhiten]

With ``prewhiten`` the model is compiled on a library whitened in place: an update then re-whitens all 62.9 GB of rows
(``beatamd_whiten_rows_batch`` with M = W_new . inv(W_old)) -- the stage-boundary cost of update_covariances on the
pre-whitened path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import beat_amd  # noqa: E402
from beat_amd.covariance import NoiseCovarianceUpdate  # noqa: E402
from beat_amd.synthetic import SyntheticSpec, build_problem, draw_population  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = beat_amd.get_context(0)
ctx.use_torch_stream()
spec = SyntheticSpec((20,), (20,), (1.0,), T=64, N=4096, D=3, S=25, time_bounds=(0.0, 0.0), covariance="toeplitz")
t0 = time.perf_counter()
prob, host = build_problem(spec, device_library=True, ctx=ctx)
PW = len(sys.argv) > 2 and sys.argv[2] == "prewhiten"
f = prob.compile(ctx, prewhiten="inplace" if PW else False)
torch.cuda.synchronize()
print("problem with dense weights built in %.1f s" % (time.perf_counter() - t0))
Q = torch.from_numpy(draw_population(spec, host["layout"], host["lower"], host["upper"], C)).cuda()
L = f.batch(Q)
upd = NoiseCovarianceUpdate(f)
q_map = Q[int(torch.argmax(L[:, -1]))].cpu().numpy()
for rep in range(3):
    ctx.enable_timing(True)
    ctx.reset_timing()
    upd.update_weights(q_map)
    torch.cuda.synchronize()
    wms, wn = ctx.kernel_time("whiten")
    parts = ["%s %.1f ms (%d)" % ((nm,) + ctx.kernel_time(nm)) for nm in ("chol_inverse", "triu_ratio", "triu_solve", "gfstack",
                                                                        "stage") if ctx.kernel_time(nm)[1]]
    ctx.enable_timing(False)
    print("  device timers: " + ", ".join(parts))
    if wn:
        flops = 64 * 400 * 3 * 25 * 4096.0 * 4096.0     # rows x N x N (upper-triangular operators: half of 2 N^2)
        print("  re-whitening: %d launches, %.1f ms, %.1f TFLOP/s" % (wn, wms, flops / (wms * 1e-3) / 1e12))
    t1 = time.perf_counter()
    L2 = f.batch(Q)
    torch.cuda.synchronize()
    print("update_weights %.1f ms (%d of 64 covariances repaired on the host), re-evaluation of %d end points %.1f ms"
          % (upd.last_ms, upd.n_repaired, C, (time.perf_counter() - t1) * 1e3))
assert bool(torch.isfinite(L2).all())
