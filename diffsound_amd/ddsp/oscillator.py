"""Damped-oscillator modules - host-side mirror of reference src/ddsp/oscillator.py.

Class names, constructor arguments, ``forward`` signatures, returned shapes/dtypes and the
``damped_freq`` attribute follow the reference; the (A, m, S) signal path itself is ONE fused HIP
kernel (``ds_osc_bank_fwd`` / ``ds_osc_bank_bwd``) wrapped in a ``torch.autograd.Function``.
The tiny per-mode parameter algebra (softplus-weighted bins, d = (alpha + beta w0^2)/2,
w = sqrt(w0^2 - d^2)) stays in torch so autograd reaches every learnable leaf exactly as in the
reference.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _hip
from ..diffelastic.material_model import Material, MatSet  # noqa: F401  (re-exported like the reference)


def modifed_sigmoid(x):
    """reference src/ddsp/utils.py:6-9 (spelling kept)."""
    return 2 * (torch.sigmoid(x) ** 2.3) + 1e-6


class WeightedParam(nn.Module):
    """Scalar = sum_j v_j softplus(p_j) / sum softplus(p)  (reference oscillator.py:10-21)."""

    def __init__(self, values_list: torch.Tensor):
        super().__init__()
        self.values_list = values_list
        self.probablity = nn.Parameter(torch.zeros(len(values_list)))
        self.probablity.data.uniform_(-1, 1)

    def forward(self):
        p = F.softplus(self.probablity)
        p = p / p.sum()
        return (self.values_list.to(p.device) * p).sum()


class WeightedSum(nn.Module):
    """reference oscillator.py:23-35."""

    def __init__(self, dims: list, vlist: list):
        super().__init__()
        # (a plain attribute in the reference: not part of the state_dict, so checkpoints load strictly both ways)
        self.register_buffer("values_list", torch.tensor([float(v) for v in vlist], dtype=torch.float32),
                             persistent=False)
        self.params = nn.Parameter(torch.zeros(*dims, len(self.values_list)))
        self.params.data.uniform_(-4, 4)

    def forward(self):
        x = F.softplus(self.params)
        x = x / x.sum(dim=-1).unsqueeze(-1)
        return (self.values_list * x).sum(dim=-1)


class DirectValue(nn.Module):
    """reference oscillator.py:38-46."""

    def __init__(self, dims: list):
        super().__init__()
        self.value = nn.Parameter(torch.zeros(*dims))
        self.value.data.uniform_(0, 0.04)

    def forward(self):
        return modifed_sigmoid(self.value)


class FilteredNoise(nn.Module):
    """Time-varying filtered white noise (DDSP style): per 64-sample frame a learnable magnitude response
    (65 bins) is turned into a Hann-windowed linear-phase FIR, applied to a fresh white-noise frame by FFT
    convolution, and the frames are overlap-added.  Interface of reference src/ddsp/filtered_noise.py:7-67
    (``coefficient_bank`` parameter, ``forward() -> (noise_num, sample_num)``); plain torch.fft plumbing, the
    noise branch is outside the hot path."""

    def __init__(self, noise_num, sample_num, filter_coeff_length=65, frame_length=64, attenuate_gain=1.0,
                 device="cuda"):
        super().__init__()
        self.noise_num, self.sample_num = noise_num, sample_num
        self.filter_coeff_length, self.frame_length, self.attenuate_gain = filter_coeff_length, frame_length, attenuate_gain
        self.coefficient_bank = nn.Parameter(torch.zeros(noise_num, sample_num // frame_length + 1, filter_coeff_length))
        self.coefficient_bank.data.uniform_(-1, 1)

    def forward(self, noise=None):
        """noise: optional (noise_num, frames, frame_length) tensor in [-1, 1) replacing the internal draw (tests)."""
        mag = modifed_sigmoid(self.coefficient_bank)
        B, nf, L = mag.shape
        taps = 2 * L - 1
        dev = mag.device
        ir = torch.fft.irfft(torch.complex(mag, torch.zeros_like(mag)), n=taps, dim=-1)  # zero-phase
        ir = torch.roll(ir, L - 1, dims=-1) * torch.hann_window(taps, dtype=torch.float32, device=dev)
        nfft = taps + self.frame_length - 1
        if noise is None:
            noise = torch.rand(B, nf, self.frame_length, device=dev) * 2 - 1
        frames = torch.fft.irfft(torch.fft.rfft(noise, n=nfft) * torch.fft.rfft(ir, n=nfft), n=nfft)
        frames = frames * self.attenuate_gain
        total = (nf - 1) * self.frame_length + nfft
        out = F.fold(frames.transpose(1, 2), output_size=(1, total), kernel_size=(1, nfft),
                     stride=(1, self.frame_length)).reshape(B, total)
        return out[:, : self.sample_num]


class _OscBank(torch.autograd.Function):
    """y[a,t] = sum_j force[a,j] sum_m amp[a,m] exp(-d_m tau) sin(w_m tau) at tau = (t-j+1)/sr."""

    @staticmethod
    def forward(ctx, d, w, amp, force, S, sr):
        _hip.require_gpu(d, w, force, amp)
        d = d.detach().double().contiguous()
        w = w.detach().double().contiguous()
        force = force.detach().float().contiguous()
        ampc = None if amp is None else amp.detach().float().contiguous()
        A, nF = force.shape
        m = d.shape[0]
        y = torch.empty((A, S), dtype=torch.float32, device=force.device)
        p = _hip.ptr
        _hip.check(_hip.lib().ds_osc_bank_fwd(p(d), p(w), p(ampc), p(force), A, m, nF, S, float(sr), p(y),
                                              _hip.stream_ptr()), "ds_osc_bank_fwd")
        ctx.save_for_backward(d, w, force, ampc if ampc is not None else torch.empty(0, device=force.device))
        ctx.has_amp = ampc is not None
        ctx.S, ctx.sr = S, float(sr)
        return y

    @staticmethod
    def backward(ctx, gy):
        d, w, force, ampc = ctx.saved_tensors
        amp = ampc if ctx.has_amp else None
        A, nF = force.shape
        m = d.shape[0]
        gy = gy.float().contiguous()
        gs = torch.empty_like(gy)
        gd = torch.empty(m, dtype=torch.float64, device=gy.device)
        gw = torch.empty(m, dtype=torch.float64, device=gy.device)
        gamp = torch.empty((A, m), dtype=torch.float32, device=gy.device) if amp is not None else None
        p = _hip.ptr
        _hip.check(_hip.lib().ds_osc_bank_bwd(p(gy), p(d), p(w), p(amp), p(force), A, m, nF, ctx.S, ctx.sr, p(gs),
                                              p(gd), p(gw), p(gamp), _hip.stream_ptr()), "ds_osc_bank_bwd")
        return gd, gw, gamp, None, None, None


class _OscBankTV(torch.autograd.Function):
    """Time-varying bank: y = FIR(sum_m amp exp(-cumsum(dmp / sr)) sin(2 pi cumsum(frq / sr))), dmp / frq (A, m, S)."""

    @staticmethod
    def forward(ctx, dmp, frq, amp, force, S, sr):
        _hip.require_gpu(dmp, frq, force, amp)
        dmp = dmp.detach().float().contiguous()
        frq = frq.detach().float().contiguous()
        force = force.detach().float().contiguous()
        ampc = None if amp is None else amp.detach().float().contiguous()
        A, m, S_ = dmp.shape
        if S_ != S or frq.shape != dmp.shape or force.shape[0] != A:
            raise ValueError("time-varying bank: dmp and frq must be (audio_num, mode_num, sample_num)")
        L = _hip.lib()
        work = torch.empty(L.ds_osc_tv_workspace_floats(A, m, S), dtype=torch.float32, device=dmp.device)
        y = torch.empty((A, S), dtype=torch.float32, device=dmp.device)
        p = _hip.ptr
        _hip.check(L.ds_osc_tv_fwd(p(dmp), p(frq), p(ampc), p(force), A, m, force.shape[1], S, float(sr), p(work), p(y),
                                   _hip.stream_ptr()), "ds_osc_tv_fwd")
        ctx.save_for_backward(dmp, frq, force, ampc if ampc is not None else torch.empty(0, device=dmp.device))
        ctx.has_amp = ampc is not None
        ctx.sr = float(sr)
        return y

    @staticmethod
    def backward(ctx, gy):
        dmp, frq, force, ampc = ctx.saved_tensors
        amp = ampc if ctx.has_amp else None
        A, m, S = dmp.shape
        gy = gy.float().contiguous()
        gs = torch.empty_like(gy)
        gd, gf = torch.empty_like(dmp), torch.empty_like(frq)
        gamp = torch.empty((A, m), dtype=torch.float32, device=gy.device) if amp is not None else None
        p = _hip.ptr
        _hip.check(_hip.lib().ds_osc_tv_bwd(p(gy), p(dmp), p(frq), p(amp), p(force), A, m, force.shape[1], S, ctx.sr,
                                            p(gs), p(gd), p(gf), p(gamp), _hip.stream_ptr()), "ds_osc_tv_bwd")
        return gd, gf, gamp, None, None, None


def oscillator_bank_tv(dmp, frq, amp, force, sample_num, sr):
    """Functional entry of the time-varying bank: dmp [1/s], frq [Hz] (A, m, S) HIP tensors (autograd ok)."""
    return _OscBankTV.apply(dmp, frq, amp, force, int(sample_num), float(sr))


def oscillator_bank(d, w, amp, force, sample_num, sr):
    """Functional entry: d, w (m,) fp64 HIP tensors (autograd ok), amp (A,m) or None, force (A,F)."""
    return _OscBank.apply(d, w, amp, force, int(sample_num), float(sr))


def _device_of(t):
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("diffsound_amd: no HIP device available (there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class _BankBase(nn.Module):
    def _setup(self, forces, audio_num, mode_num, sample_num, sr, mat):
        self.audio_num = audio_num
        self.sr = sr
        self.sample_num = sample_num
        self.mode_num = mode_num
        self.mat = mat
        self.force_frame_num = forces.shape[-1]
        self.register_buffer("_force", forces.detach().reshape(audio_num, -1).float().clone(), persistent=False)
        # the reference keeps the time-flipped conv1d weight as ``forces`` (oscillator.py:80-82)
        self.forces = torch.flip(forces.reshape(audio_num, 1, -1), [-1])

    def _render(self, freq_linear, alpha, beta, amp):
        """freq_linear (m,1) or (m,) any float dtype; alpha/beta python floats or (1,m,1) tensors; amp (A,m,1)|None."""
        dev = _device_of(self._force)
        if self._force.device != dev:
            self._force = self._force.to(dev)
        f = freq_linear.reshape(self.mode_num).to(dev).double()
        w0sq = (f * (2 * np.pi)) ** 2
        if torch.is_tensor(alpha):
            alpha = alpha.reshape(self.mode_num).to(dev).double()
        if torch.is_tensor(beta):
            beta = beta.reshape(self.mode_num).to(dev).double()
        d = 0.5 * (alpha + beta * w0sq)
        w = torch.sqrt(w0sq - d ** 2)
        fd = (w / (2 * np.pi)).float()
        self.damped_freq = fd.reshape(1, self.mode_num, 1).expand(self.audio_num, self.mode_num, self.sample_num)
        a = None if amp is None else amp.reshape(self.audio_num, self.mode_num).to(dev)
        return oscillator_bank(d, w, a, self._force, self.sample_num, self.sr)


class TraditionalDampedOscillator(_BankBase):
    """Fixed Rayleigh damping (alpha, beta from the material), unit amplitudes
    (reference oscillator.py:246-310)."""

    def __init__(self, forces, audio_num, mode_num, sample_num, sr, mat: Material):
        super().__init__()
        self._setup(forces, audio_num, mode_num, sample_num, sr, mat)
        self.alpha = mat.alpha
        self.beta = mat.beta

    def forward(self, freq_linear):
        sig = self._render(freq_linear, float(self.alpha), float(self.beta), None)
        w = self.damped_freq[0, :, 0].double() * (2 * np.pi)
        self.undamped_freq = freq_linear.reshape(1, self.mode_num, 1).to(w.device)
        return sig


class DampedOscillator(_BankBase):
    """Per-mode learnable alpha/beta (64 log-spaced bins) and per-(audio, mode) amplitude
    (reference oscillator.py:49-141)."""

    def __init__(self, forces, audio_num, mode_num, sample_num, sr, f_range: list, mat: Material):
        super().__init__()
        self._setup(forces, audio_num, mode_num, sample_num, sr, mat)
        bin_num = 64
        self.alpha_list = torch.exp(torch.linspace(np.log(mat.alpha / 10), np.log(mat.alpha * 10), bin_num))
        self.alpha = WeightedSum([1, mode_num, 1], list(self.alpha_list))
        self.beta_list = torch.exp(torch.linspace(np.log(mat.beta / 10), np.log(mat.beta * 10), bin_num))
        self.beta = WeightedSum([1, mode_num, 1], list(self.beta_list))
        self.amp = DirectValue([audio_num, mode_num, 1])
        # unused by forward(), kept so that the reference's checkpoints load strictly (oscillator.py:78: "just for load")
        self.noise = FilteredNoise(audio_num, 8000)

    def forward(self, freq_linear, non_linear_rate=0.0, noise_rate=0.0):
        return self._render(freq_linear, self.alpha(), self.beta(), self.amp())

    def _curve_render(self, freq_linear, damping_curve):
        dev = _device_of(self._force)
        if self._force.device != dev:
            self._force = self._force.to(dev)
        f = freq_linear.reshape(self.mode_num).to(dev).double()
        d = curve_on_device(damping_curve, f.detach())  # the reference evaluates the curve on detached frequencies
        w0sq = (f * (2 * np.pi)) ** 2
        w = torch.sqrt(w0sq - d ** 2)
        self.damped_freq = (w / (2 * np.pi)).float().reshape(1, self.mode_num, 1)
        return oscillator_bank(d, w, None, self._force, self.sample_num, self.sr)

    def early(self, freq_linear, damping_curve):
        """Damping per mode from ``damping_curve``, unit amplitudes, no normalisation (reference oscillator.py:85-109)."""
        return self._curve_render(freq_linear, damping_curve)

    def forward_curve(self, freq_linear, damping_curve):
        """As ``early`` with the output peak-normalised per clip (reference oscillator.py:143-176)."""
        sig = self._curve_render(freq_linear, damping_curve)
        return sig / torch.max(torch.abs(sig), dim=1, keepdim=True)[0]


def curve_on_device(curve, f):
    """damping_curve(f_i) for every mode WITHOUT the reference's per-mode host callback (oscillator.py:152-154,
    one ``interp1d`` call and two host-to-device copies per mode): a piecewise-linear table - an object with knot
    arrays ``x`` / ``y`` such as the ``scipy.interpolate.interp1d(x, y, fill_value="extrapolate")`` of
    experiments/material_real_train.py:151, or an ``(x, y)`` pair - is uploaded once and interpolated (linearly
    extrapolated beyond the end knots) on the device; any other callable is evaluated ONCE on the whole frequency
    vector (element by element only if it rejects arrays).  f: (m,) fp64 device tensor -> (m,) fp64 on f.device."""
    table = None
    if isinstance(curve, (tuple, list)) and len(curve) == 2:
        table = curve
    elif hasattr(curve, "x") and hasattr(curve, "y") and getattr(curve, "_kind", "linear") in ("linear", 1):
        table = (curve.x, curve.y)
    if table is not None:
        x = torch.as_tensor(np.asarray(table[0], dtype=np.float64), device=f.device)
        y = torch.as_tensor(np.asarray(table[1], dtype=np.float64), device=f.device)
        if x.numel() == 1:
            return y.expand_as(f).clone()
        o = torch.argsort(x)
        x, y = x[o], y[o]
        i = torch.searchsorted(x, f.contiguous()).clamp_(1, x.numel() - 1)
        x0, x1, y0, y1 = x[i - 1], x[i], y[i - 1], y[i]
        return y0 + (y1 - y0) * (f - x0) / (x1 - x0)
    fr = f.detach().cpu().numpy()
    try:
        vals = np.asarray(curve(fr), dtype=np.float64).reshape(-1)
        if vals.shape != fr.shape:
            raise ValueError
    except Exception:
        vals = np.array([float(curve(fi)) for fi in fr], dtype=np.float64)
    return torch.from_numpy(vals).to(f.device)


def init_damps(osc):
    """Pre-fit alpha/beta bins to the material table values (reference oscillator.py:314-325)."""
    optimizer = torch.optim.Adam(list(osc.alpha.parameters()) + list(osc.beta.parameters()), lr=0.01)
    for _ in range(2000):
        optimizer.zero_grad()
        loss = (osc.alpha() - osc.mat.alpha) ** 2 / osc.mat.alpha ** 2 + (osc.beta() - osc.mat.beta) ** 2 / osc.mat.beta ** 2
        loss.mean().backward()
        optimizer.step()


class GTDampedOscillator(_BankBase):
    """Free-running bank with learnable frequencies (softplus-weighted over ``f_range``), dampings and
    amplitudes, used to pre-fit recorded audio (reference oscillator.py:178-243,
    experiments/material_real_train.py:113-134).  ``freq_nonlinear`` is the reference's per-sample frequency
    offset, an (A, m, S, len(f_range)) parameter (:186-187, kept so that checkpoints and optimisers see the same
    parameter set); with ``non_linear_rate = 0`` - what the reference's callers pass - it is not touched and the
    closed-form bank renders the clip, otherwise the time-varying kernel (ds_osc_tv_fwd / ds_osc_tv_bwd) does."""

    def __init__(self, forces, audio_num, mode_num, sample_num, sr, f_range: list, mat: Material):
        super().__init__()
        self._setup(forces, audio_num, mode_num, sample_num, sr, mat)
        self.freq_linear = WeightedSum([1, mode_num, 1], f_range)
        self.freq_nonlinear = WeightedSum([audio_num, mode_num, sample_num], f_range)
        bin_num = 64
        self.alpha_list = torch.exp(torch.linspace(np.log(mat.alpha / 10), np.log(mat.alpha * 100), bin_num))
        self.alpha = WeightedSum([1, mode_num, 1], list(self.alpha_list))
        self.beta_list = torch.exp(torch.linspace(np.log(mat.beta / 10), np.log(mat.beta * 100), bin_num))
        self.beta = WeightedSum([1, mode_num, 1], list(self.beta_list))
        self.amp = DirectValue([audio_num, mode_num, 1])
        self.noise = FilteredNoise(audio_num, sample_num)

    def damping(self):
        lbd = (self.freq_linear() * 2 * np.pi) ** 2
        return 0.5 * (self.alpha() + self.beta() * lbd)

    def forward(self, non_linear_rate=0.0, noise_rate=0.0):
        if non_linear_rate != 0.0:
            sig = self._render_time_varying(non_linear_rate)
        else:
            sig = self._render(self.freq_linear(), self.alpha(), self.beta(), self.amp())
            fd = self.damped_freq[0, :, 0].double()
            d = self.damping().reshape(-1).double().to(fd.device)
            self.undamped_freq = (torch.sqrt((2 * np.pi * fd) ** 2 + d ** 2) / (2 * np.pi)).float().reshape(1, -1, 1)
        if noise_rate != 0.0:
            sig = sig + self.noise() * noise_rate
        return sig

    def _render_time_varying(self, rate):
        """reference :219-242 with the (A, m, S) cumsum / exp / sin / mode-sum / conv1d chain as one kernel pair."""
        dev = _device_of(self._force)
        if self._force.device != dev:
            self._force = self._force.to(dev)
        undamped = (self.freq_linear() + rate * self.freq_nonlinear()).to(dev)  # (A, m, S)
        lbd = (undamped * 2 * np.pi) ** 2
        damp = 0.5 * (self.alpha().to(dev) + self.beta().to(dev) * lbd)
        freq = (lbd - damp ** 2) ** 0.5 / (2 * np.pi)
        self.undamped_freq = ((2 * np.pi * freq) ** 2 + damp ** 2) ** 0.5 / (2 * np.pi)
        amp = self.amp().reshape(self.audio_num, self.mode_num).to(dev)
        return oscillator_bank_tv(damp, freq, amp, self._force, self.sample_num, self.sr)
