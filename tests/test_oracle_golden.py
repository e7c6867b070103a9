"""Pin the CPU oracle against outputs of the reference itself (fixtures made by
tests/golden/make_golden.py).  CPU only; no GPU, no /root/reference needed."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import fem, modal
from oracle import oscillator as oosc


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


# ---------------------------------------------------------------- G1 constants
@pytest.mark.parametrize("order", [1, 2])
def test_constants_bitwise(golden, order):
    g = golden("g1_constants.npz")
    pts, w = fem.gauss_points_weights(order + 2)
    assert np.array_equal(pts, g[f"gauss_pts_o{order}"])
    assert np.array_equal(w, g[f"gauss_w_o{order}"])
    L = torch.from_numpy(pts)
    assert np.array_equal(fem.shape_functions(L, order).numpy(), g[f"N_o{order}"])
    assert np.array_equal(fem.shape_function_grads(L, order).numpy(), g[f"dN_dL_o{order}"])
    assert np.array_equal(fem.element_mass_flat(order).numpy(), g[f"elem_mass_o{order}"])


def test_known_answers():
    # ord-1 exact element mass (1+delta_ab)/120 and sum of Gauss weights = 1/6
    M = fem.element_mass_table(1).numpy()
    assert abs(M[0, 0] - 1 / 60) < 1e-8 and abs(M[0, 1] - 1 / 120) < 1e-8
    for o in (1, 2):
        assert abs(fem.gauss_points_weights(o + 2)[1].sum() - 1 / 6) < 1e-7


# ---------------------------------------------------------------- G2 per-stage on the 2^3 cube
@pytest.mark.parametrize("order", [1, 2])
def test_cube_stages(golden, order):
    g = golden("g2_cube2.npz")
    rho, E, nu = g["mat"][:3]
    v, t = fem.to_high_order(torch.from_numpy(g["verts"]), torch.from_numpy(g["tets"]), order)
    assert np.array_equal(v.numpy(), g[f"o{order}_vertices"])
    assert np.array_equal(t.numpy(), g[f"o{order}_tets"])
    d = fem.OracleDeform(v, t, order)
    assert np.array_equal(d.A.numpy(), g[f"o{order}_transform"])
    sfd = d.shape_func_deriv().numpy()
    assert rel(sfd[: 4 * d.G], g[f"o{order}_sfd_first4tets"]) < 1e-6
    assert rel(d.integration_weights().numpy(), g[f"o{order}_intw"]) < 1e-6
    lam, mu = fem.lame(E, nu)
    assert rel(fem.piola_jacobian(lam, mu), g[f"o{order}_jacF"]) < 1e-12
    Kf = fem.assemble_stiffness_faithful(d, lam, mu).toarray()
    Ke = fem.assemble_stiffness(d, lam, mu).toarray()
    assert rel(Kf, g[f"o{order}_K"]) < 1e-6  # fp32 inverse differs in the last ulp across torch builds
    assert rel(Ke, Kf) < 1e-13
    M3, Ms = fem.assemble_mass(v, t, order, rho)
    assert rel(M3.toarray(), g[f"o{order}_M"]) < 1e-12


def test_rigid_modes_in_nullspace(golden):
    g = golden("g2_cube2.npz")
    K = g["o2_K"]
    x = g["o2_vertices"].astype(np.float64)
    Y = np.zeros((K.shape[0], 6))
    for c in range(3):
        Y[c::3, c] = 1
    Y[0::3, 3], Y[1::3, 3] = -x[:, 1], x[:, 0]
    Y[1::3, 4], Y[2::3, 4] = -x[:, 2], x[:, 1]
    Y[2::3, 5], Y[0::3, 5] = -x[:, 0], x[:, 2]
    assert np.abs(K @ Y).max() / np.abs(K).max() < 1e-6


# ---------------------------------------------------------------- G3 bowl
@pytest.fixture(scope="module")
def bowl1(golden):
    g = golden("g3_bowl_o1.npz")
    m = golden("g0_bowl_mesh.npz")
    rho, E, nu = g["mat"][:3]
    v = torch.from_numpy(m["verts"])
    t = torch.from_numpy(m["tets"]).long()
    d = fem.OracleDeform(v, t, 1)
    lam, mu = fem.lame(E, nu)
    K = fem.assemble_stiffness(d, lam, mu)
    M3, Ms = fem.assemble_mass(v, t, 1, rho)
    return g, d, K, M3, Ms


def test_bowl_matrices(bowl1):
    g, d, K, M3, Ms = bowl1
    assert K.shape[0] == int(g["n"])
    Kz = K.copy()
    assert Kz.nnz == int(g["nnz_K"])
    assert np.array_equal(np.diff(Kz.indptr), g["rowcnt_K"])
    assert rel(K.diagonal(), g["diag_K"]) < 2e-6
    assert rel(M3.diagonal(), g["diag_M"]) < 1e-12
    assert abs(np.sqrt((K.data ** 2).sum()) / g["fro_K"] - 1) < 1e-6
    assert abs(M3.sum() / g["sum_M"] - 1) < 1e-12
    assert rel(K @ g["x_probe"], g["Kx"]) < 2e-6
    assert rel(M3 @ g["x_probe"], g["Mx"]) < 1e-12


def test_bowl_eigen_and_readout(bowl1):
    g, d, K, M3, Ms = bowl1
    mode_num = int(g["mode_num"])
    ev, U, S, Uf = modal.eigsh_shift_invert(K, M3, mode_num)
    assert rel(ev, g["eigenvalues"]) < 1e-6
    assert rel(modal.undamped_freqs_gt(ev).numpy(), g["gt_freqs"]) < 1e-6
    assert rel(modal.get_vals(K, M3, ev, U).numpy(), g["get_vals"]) < 1e-6
    # first frequencies quoted in BASELINE.md §2
    f = modal.undamped_freqs_gt(ev).numpy().reshape(-1)
    assert np.allclose(f[:4], [2534.993, 2548.545, 6231.536, 6281.870], rtol=2e-6)


@pytest.mark.parametrize("task", ["material", "mat_baseline"])
def test_bowl_material_gradients(bowl1, task):
    g, d, K, M3, Ms = bowl1
    rho, E0, nu0 = g["mat"][:3]
    ylist = torch.from_numpy(g[f"{task}_youngs_list"])
    plist = torch.from_numpy(g[f"{task}_poisson_list"])
    yl, pl = modal.trainable_bins(E0, nu0, baseline=(task == "mat_baseline"))
    assert torch.allclose(yl, ylist) and torch.allclose(pl, plist)
    ylog = torch.from_numpy(g[f"{task}_youngs_logits"]).requires_grad_(True)
    plog = torch.from_numpy(g[f"{task}_poisson_logits"]).requires_grad_(True)
    E = modal.weighted_param(ylist, ylog)
    nu = modal.weighted_param(plist, plog)
    assert abs(float(E) / float(g[f"{task}_youngs"]) - 1) < 1e-6
    assert abs(float(nu) / float(g[f"{task}_poisson"]) - 1) < 1e-6
    lam, mu = fem.lame(float(E), float(nu))
    Kt = fem.assemble_stiffness(d, lam, mu)
    ev, U, _, _ = modal.eigsh_shift_invert(Kt, M3, int(g["mode_num"]))
    assert rel(ev, g[f"{task}_eigenvalues"]) < 1e-6
    f = modal.undamped_freqs_material(d, M3, ev, U, E, nu)
    assert rel(f.detach().numpy(), g[f"{task}_freqs"]) < 5e-6
    f.sum().backward()
    # the fp32 bracket makes the reference's own gradient noisy at the 1e-4 level
    assert rel(ylog.grad.numpy(), g[f"{task}_grad_youngs_logits"]) < 2e-3
    if task == "material":
        assert rel(plog.grad.numpy(), g[f"{task}_grad_poisson_logits"]) < 2e-3
    # closed-form gradient against the reference's autograd
    Klam = fem.assemble_stiffness(d, 1.0, 0.0)
    Kmu = fem.assemble_stiffness(d, 0.0, 1.0)
    dfdE, dfdnu = modal.closed_form_freq_grads(Klam, Kmu, f.detach().numpy(), U, float(E), float(nu))
    ylog2 = torch.from_numpy(g[f"{task}_youngs_logits"]).requires_grad_(True)
    plog2 = torch.from_numpy(g[f"{task}_poisson_logits"]).requires_grad_(True)
    E2 = modal.weighted_param(ylist, ylog2)
    nu2 = modal.weighted_param(plist, plog2)
    (E2 * dfdE.sum() + nu2 * dfdnu.sum()).backward()
    assert rel(ylog2.grad.numpy(), g[f"{task}_grad_youngs_logits"]) < 2e-3
    if task == "material":
        assert rel(plog2.grad.numpy(), g[f"{task}_grad_poisson_logits"]) < 2e-3


def test_bowl_ord2(golden):
    g = golden("g3_bowl_o2.npz")
    m = golden("g0_bowl_mesh.npz")
    rho, E, nu = g["mat"][:3]
    v, t = fem.to_high_order(torch.from_numpy(m["verts"]), torch.from_numpy(m["tets"]).long(), 2)
    assert np.array_equal(v.numpy(), g["o2_vertices"])
    assert np.array_equal(t.numpy(), g["o2_tets"])
    d = fem.OracleDeform(v, t, 2)
    lam, mu = fem.lame(E, nu)
    K = fem.assemble_stiffness(d, lam, mu)
    M3, Ms = fem.assemble_mass(v, t, 2, rho)
    assert K.nnz == int(g["nnz_K"]) == 3674016
    assert rel(K.diagonal(), g["diag_K"]) < 2e-6
    assert rel(K @ g["x_probe"], g["Kx"]) < 2e-6
    assert rel(M3 @ g["x_probe"], g["Mx"]) < 1e-12
    # mass conservation: sum(M)/3 = rho * volume
    vol = fem.tet_abs_det_f64(v, t, 2).sum().item() / 6
    assert abs(M3.sum() / 3 / (rho * vol) - 1) < 1e-6


# ---------------------------------------------------------------- G5 oscillator
def test_oscillator_traditional(golden):
    g = golden("g5_oscillator.npz")
    rho, E, nu, alpha, beta = g["mat"]
    for name in ("impulse", "random"):
        f = torch.from_numpy(g["freqs"]).clone().requires_grad_(True)
        sig, dfreq = oosc.bank(f, torch.from_numpy(g[f"trad_{name}_force"]), 8000, 32000, alpha, beta)
        assert rel(sig.detach().numpy(), g[f"trad_{name}_signal"]) < 1e-5
        assert rel(dfreq[:, :, 0].detach().numpy(), g[f"trad_{name}_damped_freq"]) < 1e-6
        (sig ** 2).mean().backward()
        assert rel(f.grad.numpy(), g[f"trad_{name}_grad_f"]) < 1e-4
        # fp64 closed form vs the fp32 cumsum path: states the audio tolerance (rel-L2 1e-3)
        cf = oosc.bank_closed_form_f64(g["freqs"], g[f"trad_{name}_force"], 8000, 32000, alpha, beta)
        err = np.linalg.norm(cf - g[f"trad_{name}_signal"]) / np.linalg.norm(cf)
        assert err < 1e-3


def test_oscillator_damped(golden):
    g = golden("g5_oscillator.npz")
    f = torch.from_numpy(g["freqs"]).clone().requires_grad_(True)
    ap = torch.from_numpy(g["damped_alpha_params"]).requires_grad_(True)
    bp = torch.from_numpy(g["damped_beta_params"]).requires_grad_(True)
    av = torch.from_numpy(g["damped_amp_value"]).requires_grad_(True)
    alpha = oosc.weighted_sum(torch.from_numpy(g["damped_alpha_values"]), ap)
    beta = oosc.weighted_sum(torch.from_numpy(g["damped_beta_values"]), bp)
    amp = oosc.modified_sigmoid(av)
    assert rel(alpha.detach().numpy(), g["damped_alpha"]) < 1e-6
    assert rel(amp.detach().numpy(), g["damped_amp"]) < 1e-6
    sig, _ = oosc.bank(f, torch.from_numpy(g["damped_forces"]), 8000, 32000, alpha, beta, amp)
    assert rel(sig.detach().numpy(), g["damped_signal"]) < 1e-5
    (sig ** 2).mean().backward()
    assert rel(f.grad.numpy(), g["damped_grad_f"]) < 1e-4
    assert rel(ap.grad.numpy(), g["damped_grad_alpha_params"]) < 1e-4
    assert rel(bp.grad.numpy(), g["damped_grad_beta_params"]) < 1e-4
    assert rel(av.grad.numpy(), g["damped_grad_amp_value"]) < 1e-4


# ---------------------------------------------------------------- G7 real-audio front half (row f4)
def _g7_leaves(g):
    A, m, S = int(g["A"]), int(g["m"]), int(g["S"])
    nl = np.random.default_rng(int(g["nl_seed"])).uniform(-4, 4, size=(A, m, S, len(g["f_range"]))).astype(np.float32)
    leaf = {k: torch.from_numpy(g[f"gt_{k}_params"]).clone().requires_grad_(True) for k in ("freq_linear", "alpha", "beta")}
    leaf["freq_nonlinear"] = torch.from_numpy(nl).requires_grad_(True)
    leaf["amp"] = torch.from_numpy(g["gt_amp_value"]).clone().requires_grad_(True)
    leaf["noise_bank"] = torch.from_numpy(g["gt_noise_bank"]).clone().requires_grad_(True)
    return leaf


@pytest.mark.parametrize("tag,rate,nrate", [("lin", 0.0, 0.0), ("tv", 0.3, 0.0), ("tvn", 0.3, 1.0)])
def test_gt_oscillator_restatement_against_reference(golden, tag, rate, nrate):
    """oracle.bank_time_varying (+ filtered_noise) against GTDampedOscillator.forward of the imported reference
    (oscillator.py:217-243): signal, undamped_freq read-out, gradients of mean(sig^2) w.r.t. every parameter."""
    g = golden("g7_real_audio.npz")
    A, m, S, sr = int(g["A"]), int(g["m"]), int(g["S"]), int(g["sr"])
    leaf = _g7_leaves(g)
    fr = torch.from_numpy(g["f_range"]).float()
    fl = oosc.weighted_sum(fr, leaf["freq_linear"])
    fnl = oosc.weighted_sum(fr, leaf["freq_nonlinear"])
    alpha = oosc.weighted_sum(torch.from_numpy(g["gt_alpha_values"]), leaf["alpha"])
    beta = oosc.weighted_sum(torch.from_numpy(g["gt_beta_values"]), leaf["beta"])
    assert rel(fl.detach().numpy(), g["gt_freq_linear"]) < 1e-6
    assert rel((0.5 * (alpha + beta * (fl * 2 * np.pi) ** 2)).detach().numpy(), g["gt_damping"]) < 1e-6
    forces = torch.from_numpy(g["forces"])
    sig, und = oosc.bank_time_varying(fl, fnl, rate, alpha, beta, oosc.modified_sigmoid(leaf["amp"]), forces, S, sr)
    if nrate:
        # FilteredNoise with the draw the reference made; torch twin of oracle.filtered_noise for the gradient
        nz = oosc.filtered_noise(leaf["noise_bank"].detach().numpy(), g[f"gt_{tag}_noise"], S)
        sig = sig + torch.from_numpy(nz).float() * nrate
    assert rel(sig.detach().numpy(), g[f"gt_{tag}_signal"]) < 2e-5
    assert rel(und.detach().mean(-1).numpy(), g[f"gt_{tag}_undamped_freq_mean"]) < 1e-6
    (sig ** 2).mean().backward()
    for k in ("freq_linear", "alpha", "beta"):
        assert rel(leaf[k].grad.numpy(), g[f"gt_{tag}_grad_{k}"]) < 2e-4, k
    assert rel(leaf["amp"].grad.numpy(), g[f"gt_{tag}_grad_amp"]) < 2e-4
    if rate:
        gnl = leaf["freq_nonlinear"].grad
        assert rel(gnl[:, :, ::40, :].numpy(), g[f"gt_{tag}_grad_nl_sample"]) < 2e-4
        assert rel(gnl.double().sum(2).numpy(), g[f"gt_{tag}_grad_nl_tsum"]) < 2e-4


def test_filtered_noise_restatement_against_reference(golden):
    g = golden("g7_real_audio.npz")
    out = oosc.filtered_noise(g["fn_bank"], g["fn_noise"], 8000)
    assert out.shape == g["fn_signal"].shape
    assert rel(out, g["fn_signal"]) < 2e-6
    # the noise branch inside GTDampedOscillator.forward(0.3, 1.0): signal difference to the noise-free render
    nz = oosc.filtered_noise(g["gt_noise_bank"], g["gt_tvn_noise"], int(g["S"]))
    assert rel(nz, g["gt_tvn_signal"] - g["gt_tv_signal"]) < 2e-5


@pytest.mark.parametrize("tag", ["early", "curve"])
def test_curve_bank_restatement_against_reference(golden, tag):
    """DampedOscillator.early / forward_curve (oscillator.py:85-109, 143-176) with the piecewise-linear damping table."""
    g = golden("g7_real_audio.npz")
    f = torch.from_numpy(g["curve_freqs"]).clone().requires_grad_(True)
    d = np.interp(g["curve_freqs"].reshape(-1).astype(np.float64), g["curve_x"], g["curve_y"])  # inside the knots
    sig, dfreq = oosc.bank_curve(f, d, torch.from_numpy(g["forces"]), int(g["S"]), int(g["sr"]), tag == "curve")
    assert rel(sig.detach().numpy(), g[f"{tag}_signal"]) < 2e-5
    assert rel(dfreq.detach().numpy(), g[f"{tag}_damped_freq"]) < 1e-6
    (sig ** 2).mean().backward()
    assert rel(f.grad.numpy(), g[f"{tag}_grad_f"]) < 2e-4
