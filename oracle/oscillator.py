"""Oracle restatement of the damped-oscillator bank (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/ddsp/oscillator.py:
  TraditionalDampedOscillator.forward :282-310
  DampedOscillator.forward            :113-141
  DampedOscillator.early / forward_curve :85-109, :143-176
  GTDampedOscillator.forward          :217-243
  WeightedSum / DirectValue           :23-46   and  src/ddsp/utils.py:6-9
  FilteredNoise.forward               src/ddsp/filtered_noise.py:20-67
Pinned by tests/golden/g5_oscillator.npz and g7_real_audio.npz (outputs of the imported reference,
tests/golden/make_golden.py) in tests/test_oracle_golden.py.
"""
import numpy as np
import torch
import torch.nn.functional as F


def modified_sigmoid(x):
    """2*sigmoid(x)^2.3 + 1e-6 (ddsp/utils.py:6-9)."""
    return 2 * (torch.sigmoid(x) ** 2.3) + 1e-6


def weighted_sum(values, params):
    """softplus-normalised combination along the last axis (oscillator.py:31-35)."""
    x = F.softplus(params)
    x = x / x.sum(dim=-1).unsqueeze(-1)
    return (values * x).sum(dim=-1)


def bank(freq, forces, sample_num, sr, alpha, beta, amp=None):
    """Reference signal path, fp32 torch, autograd-capable.

    freq (m,1) ; forces (A,F) un-flipped ; alpha, beta scalars or (1,m,1) ; amp None or (A,m,1).
    Phase and decay are accumulated with fp32 cumsum exactly like the reference (:297-298),
    then the mode sum is FIR-filtered with the force by a grouped conv1d (:306-309).
    Returns (signal (A,S), damped_freq (A,m,S))."""
    A, nF = forces.shape
    m = freq.shape[0]
    f = torch.reshape(freq, (1, m, 1)).repeat((A, 1, sample_num))
    lbd = (f * 2 * np.pi) ** 2
    damp = 0.5 * (alpha + beta * lbd)
    fd = (lbd - damp ** 2) ** 0.5 / (2 * np.pi)
    damped_freq = fd
    damp = torch.cumsum(damp / sr, dim=2)
    ph = torch.cumsum(fd / sr, dim=2)
    sig = torch.exp(-damp) * torch.sin(2 * np.pi * ph)
    if amp is not None:
        sig = amp * sig
    sig = sig.sum(1).unsqueeze(0)
    w = torch.flip(forces.reshape(A, 1, -1), [-1])
    sig = F.conv1d(sig, w, groups=A, padding=nF - 1).squeeze(0)
    return sig[:, :sample_num], damped_freq


def bank_closed_form_f64(freq, forces, sample_num, sr, alpha, beta, amp=None):
    """fp64 closed form  s[t] = sum_m a_m exp(-d_m tau) sin(w_m tau), tau=(t+1)/sr, then causal FIR.
    The limit the fp32 cumsum path approximates; used to state tolerances."""
    f = np.asarray(freq, dtype=np.float64).reshape(-1)
    forces = np.asarray(forces, dtype=np.float64)
    A = forces.shape[0]
    al = np.broadcast_to(np.asarray(alpha, dtype=np.float64).reshape(-1), f.shape) if np.ndim(alpha) else alpha
    be = np.broadcast_to(np.asarray(beta, dtype=np.float64).reshape(-1), f.shape) if np.ndim(beta) else beta
    lbd = (2 * np.pi * f) ** 2
    d = 0.5 * (al + be * lbd)
    wd = np.sqrt(lbd - d * d)
    tau = (np.arange(sample_num) + 1.0) / sr
    modes = np.exp(-d[:, None] * tau[None]) * np.sin(wd[:, None] * tau[None])  # (m,S)
    if amp is None:
        s = np.broadcast_to(modes.sum(0), (A, sample_num))
    else:
        s = (np.asarray(amp, dtype=np.float64).reshape(A, -1, 1) * modes[None]).sum(1)
    out = np.zeros((A, sample_num))
    for a in range(A):
        out[a] = np.convolve(s[a], forces[a])[:sample_num]
    return out


def bank_time_varying(freq_linear, freq_nonlinear, rate, alpha, beta, amp, forces, sample_num, sr):
    """GTDampedOscillator.forward with a per-sample frequency offset (oscillator.py:217-243), the reference's own
    chain of torch ops: undamped = f_lin + rate f_nl (A, m, S); damp = (alpha + beta (2 pi f)^2) / 2;
    freq = sqrt((2 pi f)^2 - damp^2) / 2 pi; cumsums over time; amp exp(-D) sin(2 pi P); mode sum; grouped conv1d.
    Returns (signal (A, S), undamped_freq read-out (A, m, S))."""
    A = forces.shape[0]
    undamped = freq_linear + rate * freq_nonlinear
    lbd = (undamped * 2 * np.pi) ** 2
    damp = 0.5 * (alpha + beta * lbd)
    freq = (lbd - damp ** 2) ** 0.5 / (2 * np.pi)
    und = ((2 * np.pi * freq) ** 2 + damp ** 2) ** 0.5 / (2 * np.pi)
    D = torch.cumsum(damp / sr, dim=2)
    P = torch.cumsum(freq / sr, dim=2)
    signal = (amp * torch.exp(-D) * torch.sin(2 * np.pi * P)).sum(1).unsqueeze(0)
    w = torch.flip(forces.reshape(A, 1, -1), [-1]).to(signal.dtype)
    signal = F.conv1d(signal, w, groups=A, padding=forces.shape[-1] - 1).squeeze(0)
    return signal[:, :sample_num], und


def filtered_noise(coefficient_bank, noise, sample_num, frame_length=64, attenuate_gain=1.0):
    """FilteredNoise.forward (src/ddsp/filtered_noise.py:20-67) in NumPy with the white noise GIVEN:
    coefficient_bank (B, frames, L) logits, noise (B, frames, frame_length) in [-1, 1)."""
    x = 2 * (1 / (1 + np.exp(-np.asarray(coefficient_bank, dtype=np.float64)))) ** 2.3 + 1e-6  # modifed_sigmoid, utils.py:6-9
    B, nf, L = x.shape
    taps = 2 * L - 1
    ir = np.fft.irfft(x, n=taps, axis=-1)                       # zero-phase impulse responses (:32-35)
    ir = np.roll(ir, L - 1, axis=-1)                            # causal linear phase (:39-40)
    k = np.arange(taps)
    ir = ir * (0.5 - 0.5 * np.cos(2 * np.pi * k / taps))        # torch.hann_window default: periodic (:24-25, 41-42)
    nfft = taps + frame_length - 1
    H = np.fft.rfft(ir, n=nfft, axis=-1)                        # zero-padded by frame_length - 1 (:43-46)
    N = np.fft.rfft(np.asarray(noise, dtype=np.float64), n=nfft, axis=-1)  # zero-padded by 2 L - 2 (:51-53)
    frames = np.fft.irfft(N * H, n=nfft, axis=-1) * attenuate_gain  # (:56-58)
    out = np.zeros((B, (nf - 1) * frame_length + nfft))
    for i in range(nf):                                         # overlap-add, stride frame_length (:61-65)
        out[:, i * frame_length:i * frame_length + nfft] += frames[:, i]
    return out[:, :sample_num]


def bank_curve(freq_linear, damp_per_mode, forces, sample_num, sr, normalise):
    """DampedOscillator.early (oscillator.py:85-109; normalise False) / forward_curve (:143-176; True): the damping
    of every mode is a GIVEN number (the reference evaluates a host callback on the detached frequency, :89-93 /
    :150-154), unit amplitudes; fp32 cumsum path as in the reference.  freq_linear (m, 1) torch (autograd ok),
    damp_per_mode (m,) array.  Returns (signal (A, S), damped_freq (m,))."""
    A = forces.shape[0]
    m = freq_linear.shape[0]
    d_ = torch.as_tensor(np.asarray(damp_per_mode), dtype=torch.float32).reshape(1, m, 1)
    damp = d_.repeat(A, 1, sample_num)
    lbd = (freq_linear * 2 * np.pi) ** 2
    damped_freq = (lbd - d_ ** 2) ** 0.5 / (2 * np.pi)
    freq = (lbd - damp ** 2) ** 0.5 / (2 * np.pi)
    D = torch.cumsum(damp / sr, dim=2)
    P = torch.cumsum(freq / sr, dim=2)
    signal = (torch.exp(-D) * torch.sin(2 * np.pi * P)).sum(1).unsqueeze(0)
    w = torch.flip(forces.reshape(A, 1, -1), [-1])
    signal = F.conv1d(signal, w, groups=A, padding=forces.shape[-1] - 1).squeeze(0)[:, :sample_num]
    if normalise:
        signal = signal / torch.max(torch.abs(signal), dim=1, keepdim=True)[0]
    return signal, damped_freq.reshape(-1)
