"""Oracle restatement of the multi-scale spectral loss, types 'l1_loss' and 'rmse_loss' (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/ddsp/mss_loss.py: weighted_l1_loss :50-62, SSSLoss :69-122, MSSLoss :125-147.
The reference builds its spectrograms with torchaudio.transforms.Spectrogram(n_fft, hop_length) (:79-80),
a third-party dependency absent from this image (torchaudio==2.0.2, requirements.txt:172); its documented
defaults are restated here with NumPy: periodic Hann window of n_fft samples, frames centred by reflect-padding
n_fft // 2 samples on both sides, one-sided FFT, power 2, no normalisation.
PINNED (round 5) by G9 (tests/golden/g9_mss_loss.npz): values and gradients of the reference's OWN MSSLoss, run in the build
container with torchaudio's Spectrogram supplied to it as a restatement of torchaudio 2.0.2 and inert stand-ins for torchvision /
geomloss (tests/golden/_ref_harness.install_spectral).  What the fixture pins is every line of the reference on this path -
weights, log2, eps, alpha, hop lengths, band clipping, the sum over the scales; the Spectrogram underneath remains a restated
third-party algorithm (torch.stft semantics), which tests/test_mss_loss.py checks against torch.stft itself.
"""
import numpy as np


def spectrogram(x, n_fft, hop):
    """x (batch, samples) float64 -> (batch, n_fft // 2 + 1, frames) power spectrogram."""
    x = np.asarray(x, dtype=np.float64)
    pad = n_fft // 2
    xp = np.pad(x, ((0, 0), (pad, pad)), mode="reflect")
    nframes = 1 + (xp.shape[1] - n_fft) // hop
    k = np.arange(n_fft)
    window = 0.5 - 0.5 * np.cos(2 * np.pi * k / n_fft)  # periodic Hann
    frames = np.stack([xp[:, i * hop:i * hop + n_fft] * window for i in range(nframes)], axis=1)  # (b, frames, n_fft)
    spec = np.fft.rfft(frames, axis=2)
    return np.transpose(np.abs(spec) ** 2, (0, 2, 1))


def weighted_l1(x_pred, x_true):
    """mss_loss.py:50-62."""
    T = x_pred.shape[-1]
    w = 1 - np.linspace(1.0, 0.9, T)
    w = w / w.sum() * T
    return np.abs(x_pred[:, 1:, :] * w - x_true[:, 1:, :] * w).mean()


def sss_loss(x_pred, x_true, n_fft, alpha=1.0, overlap=0.75, eps=1e-7, type="l1_loss", scale=1.0):
    """mss_loss.py:95-122 (branches 'l1_loss' and 'rmse_loss')."""
    hop = int(n_fft * (1 - overlap))
    lt, lp = spectrogram(x_true, n_fft, hop), spectrogram(x_pred, n_fft, hop)
    if type == "l1_loss":
        return alpha * weighted_l1(np.log2(lp + eps), np.log2(lt + eps)) + weighted_l1(lp, lt)
    nb = int(lt.shape[-2] * scale)
    f = lambda s: np.log2(s[..., :nb, :] + eps) - np.log2(eps)
    return np.sqrt(((f(lp) - f(lt)) ** 2).mean())


def mss_loss(x_pred, x_true, n_ffts, **kw):
    """mss_loss.py:144-147."""
    return sum(sss_loss(x_pred, x_true, n, **kw) for n in n_ffts)
