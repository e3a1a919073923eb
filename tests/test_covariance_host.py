"""Host-side pieces of the per-stage covariance update (beat_amd.covariance): the batched running-window
rms against the arrays captured from the reference's utility.running_window_rms (tests/golden/
noise_covariance.npz) and numpy's convolution for even / odd windows and windows longer than half the trace."""
import numpy as np

from conftest import load_golden


def test_running_window_rms_batch_matches_reference_arrays():
    import torch
    from beat_amd.covariance import running_window_rms_batch
    g = load_golden("noise_covariance")
    ncase = len([k for k in g.keys() if k.endswith("_data")])
    assert ncase >= 2
    for k in range(ncase):
        d, w = g["c%d_data" % k], int(g["c%d_win" % k])
        got = running_window_rms_batch(torch.from_numpy(np.ascontiguousarray(d[None])), w)[0].numpy()
        np.testing.assert_allclose(got, g["c%d_rms_same" % k], rtol=1e-12, atol=1e-15)


def test_running_window_rms_batch_windows():
    import torch
    from beat_amd.covariance import running_window_rms_batch
    rng = np.random.default_rng(0)
    for n, w in ((10, 2), (10, 3), (64, 12), (65, 13), (33, 33), (40, 39), (7, 1)):
        x = rng.standard_normal((3, n))
        ref = np.stack([np.sqrt(np.convolve(r * r, np.ones(w) / float(w), "same")) for r in x])
        got = running_window_rms_batch(torch.from_numpy(x), w).numpy()
        assert got.shape == (3, n)
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-14)
