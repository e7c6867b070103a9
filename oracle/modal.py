"""Oracle restatement of the modal analysis + differentiable read-out (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/diffelastic/diff_model.py:
  eigen_decomposition_arpack  :335-369   (scipy eigsh shift-invert, sigma=20000, k=mode_num+6, drop 6)
  get_undamped_freqs          :371-388   (first-order perturbation with detached eigenvectors)
  get_vals                    :390-399
  WeightedParam               /root/reference/src/ddsp/oscillator.py:10-21
  TrainableLinear bins        diff_model.py:51-67
"""
import numpy as np
import scipy.sparse.linalg as spla
import torch
import torch.nn.functional as F

from . import fem

SIGMA = 20000.0  # diff_model.py:357


def eigsh_shift_invert(K, M, mode_num):
    """Lowest mode_num elastic eigenpairs: ARPACK around sigma, 6 rigid pairs dropped (diff_model.py:356-369).
    Returns (eigenvalues (mode_num,), U_hat (n, mode_num), all eigenvalues (mode_num+6,), U_full)."""
    S, U = spla.eigsh(K, M=M, k=mode_num + 6, sigma=SIGMA)
    return S[6:], U[:, 6:][:, :mode_num], S, U


def weighted_param(values, logits):
    """softplus-normalised convex combination (oscillator.py:17-21)."""
    p = F.softplus(logits)
    p = p / p.sum()
    return (values * p).sum()


def trainable_bins(E0, nu0, baseline, bin_num=16):
    """Bin centres of TrainableLinear (diff_model.py:52-66)."""
    ylist = torch.exp(torch.linspace(np.log(E0 / 10), np.log(E0 * 10), bin_num))
    if baseline:
        plist = torch.linspace(nu0, nu0, 1)
    else:
        plist = torch.linspace(0.01, 0.499, bin_num)
    return ylist, plist


def undamped_freqs_material(deform, Ms3, eigenvalues, U_hat, E, nu):
    """get_undamped_freqs for task != 'gt' (diff_model.py:371-388): fp32 bracket
    lambda + diag(U^T K(theta) U) - lambda diag(U^T M U); E, nu torch scalars (autograd flows)."""
    lam, mu = fem.lame(E, nu)
    U = torch.as_tensor(U_hat).float()
    vals = torch.as_tensor(eigenvalues).float()
    KU = fem.stiff_func(deform, lam, mu, U)
    MU = torch.from_numpy(Ms3.astype(np.float32) @ U.numpy())
    pred = torch.zeros(U.shape[1]) + torch.as_tensor(eigenvalues)  # fp32 accumulator, diff_model.py:372-373
    pred = pred + (U.T @ KU).diagonal() - vals * (U.T @ MU).diagonal()
    return (torch.sqrt(pred) / 2 / np.pi).unsqueeze(1)


def undamped_freqs_gt(eigenvalues):
    """task 'gt': sqrt(lambda)/2pi in fp32 (diff_model.py:372-373,387-388)."""
    pred = torch.zeros(len(eigenvalues)) + torch.as_tensor(eigenvalues)
    return (torch.sqrt(pred) / 2 / np.pi).unsqueeze(1)


def get_vals(K, M, eigenvalues, U_hat):
    """lambda + diag(U^T K U) - lambda diag(U^T M U), fp64 compute, fp32 out (diff_model.py:390-399)."""
    U = np.asarray(U_hat)
    add = np.einsum("ij,ij->j", U, K @ U) - eigenvalues * np.einsum("ij,ij->j", U, M @ U)
    pred = torch.zeros(len(eigenvalues)) + torch.from_numpy(eigenvalues)
    pred = pred + torch.from_numpy(add)
    return pred.unsqueeze(1)


def closed_form_freq_grads(Klam, Kmu, eigen_f, U_hat, E, nu):
    """Appendix A of SURVEY.md: df_i/dE, df_i/dnu from a_i = u^T K_lam u, b_i = u^T K_mu u."""
    U = np.asarray(U_hat)
    a = np.einsum("ij,ij->j", U, Klam @ U)
    b = np.einsum("ij,ij->j", U, Kmu @ U)
    f = np.asarray(eigen_f).reshape(-1)
    dl_dE = nu / ((1 + nu) * (1 - 2 * nu))
    dl_dnu = E * (1 + 2 * nu * nu) / ((1 + nu) ** 2 * (1 - 2 * nu) ** 2)
    dm_dE = 1 / (2 * (1 + nu))
    dm_dnu = -E / (2 * (1 + nu) ** 2)
    s = 1.0 / (8 * np.pi ** 2 * f)
    return s * (a * dl_dE + b * dm_dE), s * (a * dl_dnu + b * dm_dnu)
