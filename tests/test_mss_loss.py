"""Multi-scale spectral loss.

CPU (-m "not gpu"): the NumPy oracle against torch.stft-built spectrograms (the arithmetic torchaudio's Spectrogram
documents), the module's error behaviour (no CPU path, 'geomloss' and a missing ``type`` refused).
GPU (-m gpu): the HIP kernels (ds_stft_power / ds_spec_loss / ds_stft_power_bwd behind MSSLoss) against the oracle -
loss values at every scale of the reference experiments, d loss / d audio against torch autograd through torch.stft
(fp64) and against finite differences of the oracle, determinism.
Round 5: G9 (tests/golden/g9_mss_loss.npz) holds values and gradients of the REFERENCE's own MSSLoss, run in the build
container with torchaudio's Spectrogram supplied as a restatement of torchaudio 2.0.2 (tests/golden/_ref_harness.py): the
reference's lines - weights, log2, eps, alpha, hop, the sum over the scales - are pinned by it on the CPU (oracle) and on the
device (kernels); the Spectrogram underneath stays a restated third-party algorithm (torch.stft semantics)."""
import os
import numpy as np
import pytest
import torch

from oracle import mss_loss as omss

N_FFTS = [2048, 1024, 512, 256, 128, 64]  # the scales of the reference experiments (material_sync_train.py:123-125)
REAL_FFTS = [512, 256, 128, 64, 32]       # material_real_train.py:110


def _signals(seed=0, batch=2, S=8000, sr=32000):
    rng = np.random.default_rng(seed)
    t = np.arange(S) / sr

    def mk():
        f = rng.uniform(200, 6000, size=(batch, 8, 1))
        d = rng.uniform(20, 200, size=(batch, 8, 1))
        return (np.exp(-d * t) * np.sin(2 * np.pi * f * t)).sum(1).astype(np.float32)

    return mk(), mk()


def _torch_loss(xp, xt, n_ffts, kind, alpha=1.0, eps=1e-7, scale=1.0):
    """The same loss through torch.stft (plain PyTorch reference of the op, any dtype / device)."""
    tot = 0.0
    for n in n_ffts:
        hop = int(n * 0.25)
        win = torch.hann_window(n, periodic=True, dtype=xp.dtype, device=xp.device)
        sp = lambda x: torch.stft(x, n, hop_length=hop, window=win, center=True, pad_mode="reflect",
                                  return_complex=True).abs() ** 2
        lp, lt = sp(xp), sp(xt)
        if kind == "l1_loss":
            T = lp.shape[-1]
            w = 1 - torch.linspace(1.0, 0.9, T, dtype=torch.float64).to(xp.dtype).to(xp.device)
            w = w / w.sum() * T
            wl1 = lambda a, b: ((a[:, 1:, :] - b[:, 1:, :]) * w).abs().mean()
            tot = tot + alpha * wl1((lp + eps).log2(), (lt + eps).log2()) + wl1(lp, lt)
        else:
            nb = int(lp.shape[-2] * scale)
            tot = tot + torch.sqrt((((lp[:, :nb] + eps).log2() - (lt[:, :nb] + eps).log2()) ** 2).mean())
    return tot


@pytest.mark.parametrize("kind,scale", [("l1_loss", 1.0), ("rmse_loss", 1.0), ("rmse_loss", 0.5)])
def test_oracle_matches_torch_stft(kind, scale):
    a, b = _signals()
    ref = omss.mss_loss(a, b, N_FFTS, type=kind, scale=scale)
    got = _torch_loss(torch.from_numpy(a).double(), torch.from_numpy(b).double(), N_FFTS, kind, scale=scale)
    assert abs(float(got) / ref - 1) < 1e-9


G9 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_mss_loss.npz")
G9_CASES = [("l1", "l1_loss"), ("l1_big", "l1_loss"), ("rmse", "rmse_loss"), ("rmse_half", "rmse_loss")]


@pytest.mark.parametrize("tag,kind", G9_CASES)
def test_oracle_matches_the_reference_module(tag, kind):
    """The NumPy oracle against the reference's MSSLoss (G9, fp32 spectrograms there, fp64 here): loss values, and the gradient
    the reference's autograd returned against central differences of the oracle along it."""
    g = np.load(G9)
    xp, xt, ffts, scale = g["x_pred"], g["x_true"], g[f"{tag}_n_ffts"].tolist(), float(g[f"{tag}_scale"])
    ref = float(g[f"{tag}_loss"])
    got = omss.mss_loss(xp, xt, ffts, type=kind, scale=scale)
    assert abs(got / ref - 1) < 2e-5, (got, ref)
    gr = g[f"{tag}_grad"].astype(np.float64)
    d = gr / np.linalg.norm(gr)
    h = 1e-4
    fd = (omss.mss_loss(xp + h * d, xt, ffts, type=kind, scale=scale) - omss.mss_loss(xp - h * d, xt, ffts, type=kind, scale=scale)) / (2 * h)
    assert abs(np.linalg.norm(gr) / fd - 1) < 2e-2, (np.linalg.norm(gr), fd)


def test_refusals():
    from diffsound_amd.ddsp.mss_loss import MSSLoss, SSSLoss

    with pytest.raises(ValueError):
        SSSLoss(1024, 32000, type="l2_loss")
    # the reference's default type 'geomloss' constructs (material_sync_train.py:123, material_real_train.py:109,162)
    # and delegates to the third-party Sinkhorn solver when CALLED: without the package, ImportError naming it
    g = MSSLoss([2048, 1024], 32000)
    assert [l.loss_type for l in g.losses] == ["geomloss"] * 2 and g.n_ffts == [2048, 1024]
    assert SSSLoss(1024, 32000, type="geomloss").hop_length == 256
    m = MSSLoss([256], 32000, type="l1_loss")
    a, b = _signals(1)
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.from_numpy(a), torch.from_numpy(b))  # CPU tensors: there is no CPU path


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,scale,ffts", [("l1_loss", 1.0, N_FFTS), ("rmse_loss", 1.0, N_FFTS),
                                             ("rmse_loss", 0.5, N_FFTS), ("l1_loss", 1.0, REAL_FFTS)])
def test_kernels_match_oracle(dev, kind, scale, ffts):
    from diffsound_amd.ddsp.mss_loss import MSSLoss, stft_power

    a, b = _signals(2)
    xa, xb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    m = MSSLoss(ffts, 32000, type=kind).to(dev)
    loss = m(xa, xb, scale=scale)
    ref = omss.mss_loss(a, b, ffts, type=kind, scale=scale)
    assert abs(float(loss) / ref - 1) < 2e-5
    assert float(m(xa, xb, scale=scale)) == float(loss)  # deterministic
    if kind == "l1_loss":
        assert float(m(xa, xa)) == 0.0
    for n in ffts[:2] + ffts[-1:]:  # the spectrograms themselves
        P = stft_power(xa, n, n // 4).double().cpu().numpy()
        Pref = omss.spectrogram(a, n, n // 4)
        assert P.shape == Pref.shape
        assert np.abs(P - Pref).max() / Pref.max() < 1e-6
    # log_spec (used by the reference experiments for plots, material_real_train.py:111)
    ls = m.losses[0].log_spec(xa[0], scale).cpu().numpy()
    n0 = ffts[0]
    Pr = omss.spectrogram(a[:1], n0, n0 // 4)[:, :int((n0 // 2 + 1) * scale)]
    # (bins 80 dB below the peak carry the fp32 rounding of the frame and the twiddles: ~3e-3 in log2 units)
    assert np.abs(ls - (np.log2(Pr + 1e-7) - np.log2(1e-7))).max() < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("tag,kind", G9_CASES)
def test_kernels_match_the_reference_module(dev, tag, kind):
    """The HIP path (ds_stft_power / ds_spec_loss / ds_stft_power_bwd behind MSSLoss) against the reference's MSSLoss on the same
    audio (G9): loss value within 2e-5, d loss / d audio within the tolerance the kinks of |.| and the 1e7 slopes of log2 in empty
    bins allow between two fp32 spectrograms (same bound as against the fp64 torch.stft reference below)."""
    from diffsound_amd.ddsp.mss_loss import MSSLoss

    g = np.load(G9)
    ffts, scale = g[f"{tag}_n_ffts"].tolist(), float(g[f"{tag}_scale"])
    xt = torch.from_numpy(g["x_true"]).to(dev)
    xp = torch.from_numpy(g["x_pred"]).to(dev).requires_grad_(True)
    loss = MSSLoss(ffts, int(g["sample_rate"]), type=kind).to(dev)(xp, xt, scale=scale)
    loss.backward()
    assert abs(float(loss.detach()) / float(g[f"{tag}_loss"]) - 1) < 2e-5
    gr = torch.from_numpy(g[f"{tag}_grad"]).double()
    got = xp.grad.double().cpu()
    assert float((got - gr).norm() / gr.norm()) < (1e-2 if kind == "l1_loss" else 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["l1_loss", "rmse_loss"])
def test_gradient_matches_torch_stft_autograd(dev, kind):
    from diffsound_amd.ddsp.mss_loss import MSSLoss

    a, b = _signals(3)
    xb = torch.from_numpy(b).to(dev)
    xp = torch.from_numpy(a).to(dev).requires_grad_(True)
    m = MSSLoss(N_FFTS, 32000, type=kind).to(dev)
    loss = m(xp, xb)
    loss.backward()
    g = xp.grad.double().cpu()
    xr = torch.from_numpy(a).double().requires_grad_(True)
    ref = _torch_loss(xr, torch.from_numpy(b).double(), N_FFTS, kind)
    ref.backward()
    gr = xr.grad
    assert abs(float(loss.detach()) / float(ref.detach()) - 1) < 2e-5
    # |.| has kinks and d log2(P + eps) / dP reaches 1e7 in empty bins: a few of them flip sign between the fp32
    # spectrogram and the fp64 reference (measured 4e-3 for l1_loss, < 2e-3 for rmse_loss)
    assert float((g - gr).norm() / gr.norm()) < (1e-2 if kind == "l1_loss" else 2e-3)
    # directional finite difference of the ORACLE along the kernel's gradient
    d = (g / g.norm()).numpy()
    h = 1e-4
    fd = (omss.mss_loss(a + h * d, b, N_FFTS, type=kind) - omss.mss_loss(a - h * d, b, N_FFTS, type=kind)) / (2 * h)
    assert abs(float((g * torch.from_numpy(d)).sum()) / fd - 1) < 2e-2
    # 1-D input, batch of one
    x1 = torch.from_numpy(a[0]).to(dev).requires_grad_(True)
    m(x1, xb[0]).backward()
    assert x1.grad.shape == (a.shape[1],) and torch.isfinite(x1.grad).all()


@pytest.mark.gpu
def test_loss_consumes_the_oscillator_output_on_device(dev):
    """The loop body of the reference experiments: frequencies -> oscillator bank -> MSS loss -> backward to the
    frequencies, every stage a HIP kernel."""
    from diffsound_amd.ddsp.mss_loss import MSSLoss
    from diffsound_amd.ddsp.oscillator import TraditionalDampedOscillator
    from diffsound_amd.diffelastic.material_model import Material, MatSet

    force = torch.zeros((1, 150), device=dev)
    force[0, 0] = 1
    osc = TraditionalDampedOscillator(force, 1, 16, 8000, 32000, Material(MatSet.Ceramic))
    f0 = torch.linspace(400, 9000, 16, device=dev).reshape(-1, 1)
    target = osc(f0).detach()
    f = (f0 * 1.01).clone().requires_grad_(True)
    loss = MSSLoss([1024, 512, 256, 128, 64], 32000, type="l1_loss")(osc(f), target)
    loss.backward()
    assert float(loss) > 0 and torch.isfinite(f.grad).all() and float(f.grad.abs().max()) > 0


@pytest.mark.gpu
def test_geomloss_variant_delegates_at_call_time(dev, monkeypatch):
    """Reference mss_loss.py:104-117.  Without the third-party solver the CALL raises ImportError naming it; with a
    stand-in solver in its place (test double: squared distance of the cloud means) the plumbing is checked: the
    clips are normalised, four point clouds of the right shapes arrive, the loss is alpha * log + linear, and the
    gradient reaches the mode frequencies (the spectrogram values are detached in spec2point)."""
    import sys
    import types

    from diffsound_amd.ddsp import mss_loss as M

    a, b = _signals(5)
    xa, xb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    try:
        import geomloss  # noqa: F401
        have = True
    except ImportError:
        have = False
    if not have:
        with pytest.raises(ImportError, match="geomloss"):
            M.MSSLoss([256], 32000)(xa, xb)
    seen = []

    class SamplesLoss:
        def __init__(self, loss, p, blur):
            assert (loss, p, blur) == ("sinkhorn", 2, 0.01)

        def __call__(self, x, y):
            seen.append((tuple(x.shape), tuple(y.shape)))
            return ((x.mean(1) - y.mean(1)) ** 2).sum(-1)

    monkeypatch.setitem(sys.modules, "geomloss", types.SimpleNamespace(SamplesLoss=SamplesLoss))
    m = M.MSSLoss([256, 64], 32000, alpha=2.0)
    freq = torch.tensor([700.0, 3100.0, 9000.0], device=dev, requires_grad=True)
    loss = m(xa, xb, freq, 0.5)
    assert seen == [((2, 129, 4), (2, 129, 4)), ((2, 64, 4), (2, 64, 4)), ((2, 33, 4), (2, 33, 4)),
                    ((2, 16, 4), (2, 16, 4))]
    # the same thing by hand for one scale
    s = m.losses[0]
    na, nb = M.normlize(xa), M.normlize(xb)
    want_lin = SamplesLoss("sinkhorn", 2, 0.01)(M.spec2point(s.spec(na), freq, 32000), M.spec2point(s.spec(nb)))
    want_log = SamplesLoss("sinkhorn", 2, 0.01)(M.spec2point(s.log_spec(na, 0.5) / 40, freq, 32000),
                                                M.spec2point(s.log_spec(nb, 0.5) / 40))
    one = s(xa, xb, freq, 0.5)
    assert torch.allclose(one, 2.0 * want_log + want_lin)
    loss.backward()
    assert freq.grad is not None and float(freq.grad.abs().sum()) > 0
