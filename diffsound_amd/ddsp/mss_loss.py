"""Multi-scale spectral loss head on HIP kernels.

Interface of reference src/ddsp/mss_loss.py (``SSSLoss`` :70-122, ``MSSLoss`` :125-147: constructor arguments,
``forward(x_pred, x_true, freq=None, scale=1.0)``, ``log_spec``); the arithmetic is three kernels of
libdiffsound_hip.so (csrc/stft.hip) behind one ``torch.autograd.Function`` per scale, so the rendered audio is consumed
where the oscillator bank left it in HBM: ``ds_stft_power`` (Hann-windowed, reflect-centred STFT power - what
``torchaudio.transforms.Spectrogram(n_fft, hop_length)`` computes, reference :79-80), ``ds_spec_loss`` (the weighted
L1 on linear and log2 power, reference :50-62, 98-103, or the log RMSE, :118-122, with d loss / d P) and
``ds_stft_power_bwd`` (d loss / d audio).  Gradients reach ``x_pred`` only (the target is data).

``type`` is 'l1_loss', 'rmse_loss' (HIP kernels) or 'geomloss', the reference's default (:104-117): that variant
normalises both clips, turns their spectrograms into point clouds (``spec2point``, :19-48: the spectrogram VALUES are
detached there, so the gradient reaches the prediction only through the mode positions ``freq``) and hands them to the
third-party Sinkhorn solver ``geomloss.SamplesLoss("sinkhorn", p=2, blur=0.01)`` (``geomloss==0.2.6``,
requirements.txt:46).  It stays on PyTorch: the spectrograms come from ``ds_stft_power``,
the point clouds are torch ops, the solver is imported WHEN THE LOSS IS CALLED - constructing the module never needs
the package, calling it without the package raises ImportError naming it.
Parity (round 5): values and gradients of the reference module itself - run in the build container with torchaudio's
Spectrogram restated (torchaudio==2.0.2, absent from the image) - are the fixture G9; the kernels meet it to 2e-5
(tests/test_mss_loss.py), beside the checks against oracle/mss_loss.py and torch.stft.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import _hip

_KIND = {"l1_loss": 0, "rmse_loss": 1}
_TYPES = ("l1_loss", "rmse_loss", "geomloss")


def clip_spec(x, scale):
    """Lowest ``scale`` fraction of the frequency bins of a (batch, freq, time) spectrogram (reference :14-16)."""
    return x[..., :int(x.shape[-2] * scale), :]


def weighted_l1_loss(x_pred, x_true):
    """Mean |difference| of two (batch, freq, time) spectrograms without the DC bin, frames weighted by a ramp
    that rises from 0 and is normalised to mean 1 (reference :50-62).  Plain torch: the module-level helper the
    reference exports; ``SSSLoss(type='l1_loss')`` itself runs the fused kernel ``ds_spec_loss``."""
    T = x_pred.shape[-1]
    w = 1 - torch.linspace(1.0, 0.9, T).to(x_pred.device)
    w = (w / w.sum() * T)[None, None, :]
    return torch.nn.functional.l1_loss(x_pred[:, 1:, :] * w, x_true[:, 1:, :] * w)


def normlize(x):
    """Clip divided by its (detached) maximum + 1e-7 (reference :65-68; the spelling is the reference's)."""
    return x / (x.detach().max(-1)[0].unsqueeze(-1) + 1e-7)


def spec2point(x, freq=None, sample_rate=None):
    """(batch, freq, time) spectrogram -> point cloud (batch, freq, 4) for the Sinkhorn loss (reference :18-48):
    three features = the row's time profile resampled linearly to 3 values (DETACHED), the fourth = the row's
    relative frequency position.  With ``freq`` (mode frequencies in Hz) the positions of the bins within +-2 of
    every mode are replaced by the exact, differentiable offsets ``(bin(freq) -+ w) / bins`` - the only place a
    gradient enters."""
    bins = x.shape[-2]
    x = x.detach()
    nfeat = 3
    pts = torch.zeros(x.shape[0], x.shape[1], nfeat + 1, device=x.device)
    pts[:, :, :nfeat] = torch.nn.functional.interpolate(x, size=nfeat, mode="linear")
    pts[:, :, nfeat] = (torch.arange(bins, dtype=torch.float32, device=x.device) / bins)[None, :]
    if freq is not None:
        pos = bins / (sample_rate // 2) * freq
        for w in range(2, -1, -1):
            for cur in (pos - w, pos + w):
                row = cur.long()
                ok = (row >= 0) & (row < bins)
                pts[:, row[ok], nfeat] = cur[ok] / bins
    return pts


def _sinkhorn():
    try:
        from geomloss import SamplesLoss
    except ImportError as e:  # the package is third-party and not part of this library
        raise ImportError("SSSLoss/MSSLoss(type='geomloss') delegates to the third-party package 'geomloss' "
                          "(reference requirements.txt:46: geomloss==0.2.6), which is not installed; install it or "
                          "use type='l1_loss' / 'rmse_loss' (HIP kernels)") from e
    return SamplesLoss(loss="sinkhorn", p=2, blur=0.01)


def _as_clips(x):
    _hip.require_gpu(x)
    if x.dim() == 1:
        x = x.unsqueeze(0)
    if x.dim() != 2:
        raise ValueError("spectral loss: audio must be (batch, samples)")
    return x


def stft_power(x, n_fft, hop, keep_parts=False):
    """(B, S) f32 HIP tensor -> power spectrogram (B, n_fft // 2 + 1, 1 + S // hop) [, re, im]."""
    x = _as_clips(x).detach().float().contiguous()
    B, S = x.shape
    F_, T = n_fft // 2 + 1, 1 + S // hop
    P = torch.empty((B, F_, T), dtype=torch.float32, device=x.device)
    re = torch.empty_like(P) if keep_parts else None
    im = torch.empty_like(P) if keep_parts else None
    p = _hip.ptr
    _hip.check(_hip.lib().ds_stft_power(p(x), B, S, n_fft, hop, p(P), p(re), p(im), _hip.stream_ptr()), "ds_stft_power")
    return (P, re, im) if keep_parts else P


class _SpecLoss(torch.autograd.Function):
    """One scale: loss(x_pred, x_true) -> scalar; backward to x_pred."""

    @staticmethod
    def forward(ctx, x_pred, x_true, n_fft, hop, kind, alpha, eps, scale):
        xp = _as_clips(x_pred)
        B, S = xp.shape
        need_grad = x_pred.requires_grad
        Pp, re, im = stft_power(xp, n_fft, hop, keep_parts=True) if need_grad else (stft_power(xp, n_fft, hop), None, None)
        Pt = stft_power(x_true, n_fft, hop)
        if Pt.shape != Pp.shape:
            raise ValueError("spectral loss: prediction and target must have the same shape")
        F_, T = Pp.shape[1], Pp.shape[2]
        fclip = int(F_ * scale)
        sums = torch.empty((B, F_, 2), dtype=torch.float64, device=Pp.device)
        gP = torch.empty_like(Pp) if need_grad else None
        p = _hip.ptr
        _hip.check(_hip.lib().ds_spec_loss(kind, p(Pp), p(Pt), B, F_, T, float(alpha), float(eps), fclip, p(sums), p(gP),
                                           _hip.stream_ptr()), "ds_spec_loss")
        tot = sums.sum((0, 1))
        if kind == 0:
            loss = (alpha * tot[0] + tot[1]) / (B * (F_ - 1) * T)
            post = None
        else:
            loss = torch.sqrt(tot[0] / (B * fclip * T))
            post = loss  # d sqrt(m) = d m / (2 sqrt(m)); the kernel wrote d (m) / 2
        ctx.save_for_backward(gP, re, im, post if post is not None else torch.empty(0, device=Pp.device))
        ctx.meta = (B, S, n_fft, hop, T, x_pred.dim(), post is not None)
        return loss.float()

    @staticmethod
    def backward(ctx, g):
        gP, re, im, post = ctx.saved_tensors
        B, S, n_fft, hop, T, xdim, has_post = ctx.meta
        if gP is None:
            return (None,) * 8
        gscale = g.double() / post if has_post else g.double()
        gframes = torch.empty((B, T, n_fft), dtype=torch.float32, device=gP.device)
        gx = torch.empty((B, S), dtype=torch.float32, device=gP.device)
        p = _hip.ptr
        _hip.check(_hip.lib().ds_stft_power_bwd(p(gP), p(re), p(im), B, S, n_fft, hop, 1.0, p(gframes), p(gx),
                                                _hip.stream_ptr()), "ds_stft_power_bwd")
        gx = gx * gscale.float()  # the upstream gradient stays on the device: no host synchronisation in backward
        return (gx if xdim == 2 else gx[0]), None, None, None, None, None, None, None


class SSSLoss(nn.Module):
    """Single-scale spectral loss (reference :70-122)."""

    def __init__(self, n_fft, sample_rate, alpha=1.0, overlap=0.75, eps=1e-7, type="geomloss"):
        super().__init__()
        if type not in _TYPES:
            raise ValueError(f"SSSLoss: type={type!r}; expected one of {_TYPES}")
        self.n_fft, self.alpha, self.eps = n_fft, alpha, eps
        self.geomloss = None  # the Sinkhorn solver, created on the first 'geomloss' call
        self.hop_length = int(n_fft * (1 - overlap))
        self.loss_type = type
        self.sample_rate = sample_rate

    def spec(self, x):
        return stft_power(x, self.n_fft, self.hop_length)

    def log_func(self, x):
        return (x + self.eps).log2() - np.log2(self.eps)

    def log_spec(self, x, scale=1.0):
        S = self.spec(x)
        return self.log_func(S[..., :int(S.shape[-2] * scale), :])

    def forward(self, x_pred, x_true, freq=None, scale=1.0):
        if self.loss_type == "geomloss":
            return self._forward_geomloss(x_pred, x_true, freq, scale)
        return _SpecLoss.apply(x_pred, x_true, self.n_fft, self.hop_length, _KIND[self.loss_type], self.alpha, self.eps,
                               scale)

    def _forward_geomloss(self, x_pred, x_true, freq, scale):
        """Reference :104-117.  Spectrograms by ``ds_stft_power`` (their values carry no gradient here: spec2point
        detaches them), point clouds in torch, Sinkhorn divergence by the third-party solver."""
        if self.geomloss is None:
            self.geomloss = _sinkhorn()
        x_true, x_pred = normlize(_as_clips(x_true)), normlize(_as_clips(x_pred))
        lin_true, lin_pred = self.spec(x_true), self.spec(x_pred)
        log_true, log_pred = self.log_spec(x_true, scale) / 40, self.log_spec(x_pred, scale) / 40
        loss_lin = self.geomloss(spec2point(lin_pred, freq, self.sample_rate), spec2point(lin_true))
        loss_log = self.geomloss(spec2point(log_pred, freq, self.sample_rate), spec2point(log_true))
        return self.alpha * loss_log + loss_lin


class MSSLoss(nn.Module):
    """Multi-scale spectral loss: sum of ``SSSLoss`` over ``n_ffts`` (reference :125-147)."""

    def __init__(self, n_ffts, sample_rate, alpha=1.0, overlap=0.75, eps=1e-7, type="geomloss"):
        super().__init__()
        self.n_ffts = n_ffts
        self.losses = nn.ModuleList([SSSLoss(n_fft, sample_rate, alpha, overlap, eps, type) for n_fft in n_ffts])

    def forward(self, x_pred, x_true, freq=None, scale=1.0):
        return sum(loss(x_pred, x_true, freq, scale) for loss in self.losses).sum()
