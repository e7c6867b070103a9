"""Drop-in boundary: every name the reference's experiment scripts import from ``src.*`` on the
path resolves here, the set-up lines of those scripts construct, and the small helpers behave as the reference's.
The name list is a committed fixture generated from the reference with ``ast`` (tests/golden/make_import_list.py)."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "experiment_imports.json")))


@pytest.mark.parametrize("module", sorted(FIX["in_scope"]))
def test_experiment_import_lines_resolve(module):
    mod = importlib.import_module(module)
    for name in FIX["in_scope"][module]:
        assert hasattr(mod, name), f"from {module} import {name} (used by {FIX['in_scope'][module][name]})"


def test_fixture_covers_the_material_scripts():
    used = {s for names in FIX["in_scope"].values() for scripts in names.values() for s in scripts}
    assert {"material_sync_train.py", "material_real_train.py", "geometry_train.py"} <= used
    assert set(FIX["in_scope"]["src.utils.utils"]) == {"plot_spec", "resample"}


def test_loss_heads_of_the_material_scripts_construct():
    """material_sync_train.py:123-125, material_real_train.py:109-110,162: all three loss types construct (the
    Sinkhorn variant needs its third-party solver only when called) and expose ``losses[i].log_spec`` / ``n_ffts``."""
    from src.ddsp.mss_loss import MSSLoss, clip_spec, weighted_l1_loss

    early = MSSLoss([2048, 1024], 32000, type="geomloss")
    late = MSSLoss([1024, 512, 256, 128, 64], 32000, type="l1_loss")
    rmse = MSSLoss([1024, 512, 256, 128, 64], 32000, type="rmse_loss")
    assert MSSLoss([512], 32000).losses[0].loss_type == "geomloss"  # the reference's default
    assert callable(late.losses[0].log_spec) and late.n_ffts == [1024, 512, 256, 128, 64]
    assert len(early.losses) == 2 and len(rmse.losses) == 5
    x = torch.rand(2, 9, 5)
    assert clip_spec(x, 0.5).shape == (2, 4, 5)
    assert float(weighted_l1_loss(x, x)) == 0.0


def test_weighted_l1_and_point_clouds_follow_the_reference_formulas():
    from oracle import mss_loss as omss
    from src.ddsp.mss_loss import normlize, spec2point, weighted_l1_loss

    rng = np.random.default_rng(0)
    a, b = rng.random((2, 17, 31)), rng.random((2, 17, 31))
    got = float(weighted_l1_loss(torch.from_numpy(a), torch.from_numpy(b)))
    assert abs(got / omss.weighted_l1(a, b) - 1) < 1e-6
    x = torch.from_numpy(rng.standard_normal((3, 100)).astype(np.float32))
    n = normlize(x)
    assert torch.allclose(n, x / (x.max(-1)[0][:, None] + 1e-7))
    # point clouds: 3 linearly resampled time features + relative bin position; mode positions are differentiable
    spec = torch.from_numpy(rng.random((1, 64, 12)).astype(np.float32))
    pts = spec2point(spec)
    assert pts.shape == (1, 64, 4)
    assert torch.allclose(pts[0, :, 3], torch.arange(64.0) / 64)
    lo, hi = spec[0, :, 1:3].mean(-1), spec[0, :, 9:11].mean(-1)  # linear resampling, half-pixel centres: 1.5, 5.5, 9.5
    assert torch.allclose(pts[0, :, 0], lo, atol=1e-6) and torch.allclose(pts[0, :, 2], hi, atol=1e-6)
    freq = torch.tensor([1000.0, 5250.0], requires_grad=True)
    pf = spec2point(spec, freq, 32000)
    pos = 64 / 16000 * freq.detach()
    # the loop runs w = 2, 1, 0 and writes pos - w before pos + w: bin(pos) ends as pos, bin(pos) + 2 as pos + 2
    assert float(pf[0, int(pos[0]), 3]) == pytest.approx(float(pos[0]) / 64)
    assert float(pf[0, int(pos[1]) + 2, 3]) == pytest.approx((float(pos[1]) + 2) / 64)
    pf[0, :, 3].sum().backward()
    assert freq.grad is not None and float(freq.grad.abs().sum()) > 0


def test_resample_is_the_windowed_sinc_polyphase_filter():
    from scipy.signal import resample_poly
    from src.utils.utils import resample

    sr, new = 48000, 32000  # material_real_train.py:97
    t = np.arange(sr // 4) / sr
    x = (np.sin(2 * np.pi * 440 * t) + 0.3 * np.sin(2 * np.pi * 5000 * t + 0.3)).astype(np.float64)
    y = resample(torch.from_numpy(x)[None, :], sr, new)
    assert y.shape == (1, int(np.ceil(len(x) * 2 / 3)))
    tn = np.arange(y.shape[1]) / new
    want = np.sin(2 * np.pi * 440 * tn) + 0.3 * np.sin(2 * np.pi * 5000 * tn + 0.3)
    mid = slice(200, -200)  # away from the zero-padded ends
    assert np.abs(y[0].numpy()[mid] - want[mid]).max() < 2e-3
    ref = resample_poly(x, 2, 3)  # an independent polyphase resampler (Kaiser window): same signal to filter ripple
    assert np.abs(y[0].numpy()[mid] - ref[mid]).max() < 5e-3
    # above the new Nyquist is removed; batch shape kept; identity when the rates agree
    hi = np.sin(2 * np.pi * 20000 * t)
    assert resample(torch.from_numpy(hi)[None], sr, new)[0, 200:-200].abs().max() < 2e-2
    z = torch.from_numpy(x).float().reshape(1, 1, -1).repeat(2, 3, 1)
    assert resample(z, sr, new).shape == (2, 3, y.shape[1])
    assert resample(z, sr, sr) is z


def test_plot_spec_returns_a_figure():
    pytest.importorskip("matplotlib")
    from src.utils.utils import plot_spec

    fig = plot_spec(torch.rand(33, 20), torch.rand(33, 20))
    assert fig.__class__.__name__ == "Figure" and len(fig.axes) == 1
    img = fig.axes[0].images[0].get_array()
    assert img.shape == (33, 40)
    import matplotlib.pyplot as plt
    plt.close(fig)
