"""Multi-scale spectral loss: the torch.stft-based mirror of the reference module against
the NumPy oracle - on CPU always, on the HIP device under -m gpu - plus its gradient and error behaviour."""
import numpy as np
import pytest
import torch

from diffsound_amd.ddsp.mss_loss import MSSLoss, SSSLoss
from oracle import mss_loss as omss

N_FFTS = [2048, 1024, 512, 256, 128, 64]  # the scales of the reference experiments (material_sync_train.py:123-125)


def _signals(seed=0, batch=2, S=8000, sr=32000):
    rng = np.random.default_rng(seed)
    t = np.arange(S) / sr
    def mk():
        f = rng.uniform(200, 6000, size=(batch, 8, 1)); d = rng.uniform(20, 200, size=(batch, 8, 1))
        return (np.exp(-d * t) * np.sin(2 * np.pi * f * t)).sum(1).astype(np.float32)
    return mk(), mk()


@pytest.mark.parametrize("kind,scale", [("l1_loss", 1.0), ("rmse_loss", 1.0), ("rmse_loss", 0.5)])
def test_matches_oracle_cpu(kind, scale):
    a, b = _signals()
    loss = MSSLoss(N_FFTS, 32000, type=kind)(torch.from_numpy(a).double(), torch.from_numpy(b).double(), scale=scale)
    ref = omss.mss_loss(a, b, N_FFTS, type=kind, scale=scale)
    assert abs(float(loss) / ref - 1) < 2e-6  # the module's time weights are fp32 (torch.linspace default), as in the reference
    loss32 = MSSLoss(N_FFTS, 32000, type=kind)(torch.from_numpy(a), torch.from_numpy(b), scale=scale)
    assert abs(float(loss32) / ref - 1) < 2e-4  # fp32 spectrograms


def test_zero_for_identical_signals_and_gradient_direction():
    a, b = _signals(1)
    m = MSSLoss(N_FFTS, 32000, type="l1_loss")
    xa = torch.from_numpy(a)
    assert float(m(xa, xa)) == 0.0
    xp = torch.from_numpy(b).clone().requires_grad_(True)
    m(xp, xa).backward()
    g = xp.grad
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    # a small step against the gradient lowers the loss
    with torch.no_grad():
        l0 = float(m(xp, xa)); l1 = float(m(xp - 1e-3 * g / g.norm(), xa))
    assert l1 < l0


def test_geomloss_variant_is_refused():
    with pytest.raises(NotImplementedError):
        SSSLoss(1024, 32000, type="geomloss")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["l1_loss", "rmse_loss"])
def test_matches_oracle_on_device(kind):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    a, b = _signals(2)
    dev = torch.device("cuda:0")
    m = MSSLoss(N_FFTS, 32000, type=kind).to(dev)
    loss = m(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
    ref = omss.mss_loss(a, b, N_FFTS, type=kind)
    assert abs(float(loss) / ref - 1) < 2e-4
