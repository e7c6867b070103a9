"""Glue helpers the reference's experiment scripts import from ``src.utils.utils`` (reference src/utils/utils.py):
``LOBPCG_solver_freq`` (:80-90, on the hot path), ``resample`` (:111-113) and ``plot_spec`` (:164-173) - the two
names ``experiments/material_sync_train.py:16`` / ``material_real_train.py:16`` need before they can start.
The COMSOL / audio-folder loaders of that file are outside the path and not provided."""
import math

import torch

from ..lobpcg import lobpcg_func


def LOBPCG_solver_freq(stiff_matrix, mass_matrix, niter=1000, freq_limit=None, k=100):
    """k+6 lowest pairs of (K, M), optional frequency cut, the first six (rigid) dropped
    (reference utils.py:80-90)."""
    vals, vecs = lobpcg_func(stiff_matrix, mass_matrix, k + 6, niter=niter, tracker=None, largest=False)
    if freq_limit:
        eigenvalue_limit = (freq_limit * 2 * 3.14159) ** 2
        mask = vals < eigenvalue_limit
        vals = vals[mask]
        vecs = vecs[:, mask]
    return vals[6:], vecs[:, 6:]


def _sinc_resample_kernel(orig, new, width=6, rolloff=0.99, dtype=torch.float64):
    """Polyphase bank of Hann-windowed sinc filters for the rational rate change orig -> new: the published
    algorithm behind ``torchaudio.transforms.Resample`` (its defaults: 'sinc_interp_hann', lowpass_filter_width 6,
    rolloff 0.99), which is what reference utils.py:112 builds.  Returns (kernels (new, taps), half-width)."""
    g = math.gcd(int(orig), int(new))
    orig, new = int(orig) // g, int(new) // g
    base = min(orig, new) * rolloff
    half = int(math.ceil(width * orig / base))
    idx = torch.arange(-half, half + orig, dtype=dtype)[None, :] / orig
    phase = torch.arange(0, -new, -1, dtype=dtype)[:, None] / new
    t = ((phase + idx) * base).clamp(-width, width)
    window = torch.cos(t * math.pi / width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / torch.where(t == 0, torch.ones_like(t), t))
    return kern * window * (base / orig), half, orig, new


def resample(waveform, sample_rate, new_sample_rate):
    """``waveform (..., time)`` at ``sample_rate`` -> the same clip at ``new_sample_rate`` (reference utils.py:111-113:
    ``torchaudio.transforms.Resample(sample_rate, new_sample_rate)``).  Data preparation, runs once per clip: plain
    torch on the tensor's own device (strided conv1d with the polyphase bank), no torchaudio needed."""
    if int(sample_rate) == int(new_sample_rate):
        return waveform
    kern, half, orig, new = _sinc_resample_kernel(sample_rate, new_sample_rate)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1])
    length = x.shape[-1]
    kern = kern.to(device=x.device, dtype=x.dtype)
    x = torch.nn.functional.pad(x, (half, half + orig))
    y = torch.nn.functional.conv1d(x[:, None], kern[:, None], stride=orig)  # (batch, new, frames)
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(new * length / orig))
    return y[..., :target].reshape(shape[:-1] + (target,))


def plot_spec(spec_gt, spec_predict):
    """Side-by-side image of two (freq, time) spectrograms as a matplotlib figure (reference utils.py:164-173;
    the experiments hand it to TensorBoard).  matplotlib is imported on call."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    fig = plt.figure(figsize=(10, 5))
    img = torch.cat([spec_gt, spec_predict], dim=1)
    plt.imshow(img.detach().cpu().numpy(), origin="lower", aspect="auto", cmap="magma")
    fig.tight_layout(pad=0)
    return fig
