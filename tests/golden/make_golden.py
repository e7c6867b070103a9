#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE DiffSound code on CPU.

Run only in the build container (needs /root/reference):

    python tests/golden/make_golden.py [--skip-ord2-bowl]

Outputs small ``.npz`` fixtures next to this file.  The fixtures are data
(inputs + outputs of the reference); no reference source is stored.  The
reference entry points exercised are cited next to each block.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness  # noqa: E402

_ref_harness.install()

from src.diffelastic.diff_model import DiffSoundObj, FixedLinear, TrainableLinear, build_model  # noqa: E402
from src.diffelastic.mesh import TetMesh  # noqa: E402
from src.diffelastic.deform import Deform  # noqa: E402
from src.diffelastic.gauss import generate_gauss_points_weights  # noqa: E402
from src.diffelastic.shape_func import get_shape_function, get_shape_function_grad  # noqa: E402
from src.diffelastic.mass_matrix import get_elememt_mass_matrix  # noqa: E402
from src.diffelastic.material_model import Material  # noqa: E402
from src.ddsp.oscillator import TraditionalDampedOscillator, DampedOscillator, GTDampedOscillator  # noqa: E402
from src.ddsp.filtered_noise import FilteredNoise  # noqa: E402
from src.lobpcg import lobpcg_func  # noqa: E402

MAT = (2700.0, 5e10, 0.25, 6.0, 1e-7)
BOWL = os.path.join(_ref_harness.REFERENCE_ROOT, "data/mesh/bowl/bowl.obj")


def kuhn_cube(n, box=(0.10, 0.08, 0.06), jitter=0.15, seed=1234):
    """Structured Kuhn/Freudenthal box mesh; stored in the fixture."""
    nx = ny = nz = n
    xs = np.linspace(0, box[0], nx + 1)
    ys = np.linspace(0, box[1], ny + 1)
    zs = np.linspace(0, box[2], nz + 1)
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3)
    rng = np.random.default_rng(seed)
    h = np.array([box[0] / nx, box[1] / ny, box[2] / nz])
    interior = np.ones((nx + 1, ny + 1, nz + 1), bool)
    interior[[0, -1], :, :] = False
    interior[:, [0, -1], :] = False
    interior[:, :, [0, -1]] = False
    jit = rng.uniform(-jitter, jitter, size=verts.shape) * h
    verts = verts + jit * interior.reshape(-1, 1)

    def vid(i, j, k):
        return (i * (ny + 1) + j) * (nz + 1) + k

    tets = []
    for i in range(nx):
        for j in range(ny):
            for k in range(nz):
                c = [vid(i, j, k), vid(i + 1, j, k), vid(i + 1, j + 1, k), vid(i, j + 1, k),
                     vid(i, j, k + 1), vid(i + 1, j, k + 1), vid(i + 1, j + 1, k + 1), vid(i, j + 1, k + 1)]
                for a, b in ((1, 2), (2, 3), (3, 7), (7, 4), (4, 5), (5, 1)):
                    tets.append([c[0], c[a], c[b], c[6]])
    return verts.astype(np.float32), np.asarray(tets, dtype=np.int64)


def to_dense(sp):
    return sp.to_dense().numpy()


def constants():
    """G1: src/diffelastic/gauss.py:17-38, shape_func.py:3-108, mass_matrix.py:9-31."""
    out = {}
    for order in (1, 2):
        pts, w = generate_gauss_points_weights(order + 2)
        out[f"gauss_pts_o{order}"] = pts
        out[f"gauss_w_o{order}"] = w
        L = torch.from_numpy(pts)
        out[f"N_o{order}"] = get_shape_function(L, order).numpy()
        out[f"dN_dL_o{order}"] = get_shape_function_grad(L, order).numpy()
        out[f"elem_mass_o{order}"] = get_elememt_mass_matrix(order).numpy()
    np.savez_compressed(os.path.join(HERE, "g1_constants.npz"), **out)
    print("g1 done")


def per_stage_cube():
    """G2: per-stage tensors on a 2^3 Kuhn cube, ord 1 and 2 (mesh.py:58-160, deform.py:35-147,
    diff_model.py:184-312)."""
    verts, tets = kuhn_cube(2)
    out = {"verts": verts, "tets": tets, "mat": np.asarray(MAT)}
    for order in (1, 2):
        obj = DiffSoundObj(vertices=torch.from_numpy(verts), tets=torch.from_numpy(tets), mode_num=8,
                           mat=MAT, order=order, mat_model=FixedLinear, task="gt")
        tm = obj.tetmesh
        out[f"o{order}_vertices"] = tm.vertices.numpy()
        out[f"o{order}_tets"] = tm.tets.numpy()
        out[f"o{order}_transform"] = tm.transform_matrix.numpy()
        sfd = obj.deform.shape_func_deriv.numpy()
        G = obj.deform.num_guass_points
        out[f"o{order}_sfd_first4tets"] = sfd[: 4 * G]
        out[f"o{order}_intw"] = obj.deform.integration_weights.numpy().reshape(-1)
        obj.update_mass_matrix(MAT[0])
        obj.update_stiff_matrix()
        out[f"o{order}_K"] = to_dense(obj.stiff_matrix)
        out[f"o{order}_M"] = to_dense(obj.mass_matrix)
        out[f"o{order}_jacF"] = obj.material_model.jacobian_F().numpy().reshape(9, 9)
    np.savez_compressed(os.path.join(HERE, "g2_cube2.npz"), **out)
    print("g2 done")


def bowl(order, mode_num=32):
    """G3: bowl fixture end to end (diff_model.py:98-113,184-399)."""
    t0 = time.time()
    pts, tets = _ref_harness.read_gmsh22_binary(BOWL + "_.msh")
    out = {"mat": np.asarray(MAT), "mode_num": mode_num, "order": order}
    if order == 1:
        # G0: the mesh itself, exactly as TetMesh.from_triangle_mesh hands it over (f32 verts)
        np.savez_compressed(os.path.join(HERE, "g0_bowl_mesh.npz"), verts=pts.astype(np.float32),
                            tets=tets.astype(np.int32))

    # --- task "gt": FixedLinear, eigen decomposition, eigenvalues & gt freqs
    gt = build_model(BOWL, mode_num=mode_num, order=order, mat=MAT, task="gt")
    gt.update_mass_matrix(MAT[0])
    gt.update_stiff_matrix()
    K = gt.stiff_matrix
    M = gt.mass_matrix
    n = K.shape[0]
    out["n"] = n
    out["nnz_K"] = K._nnz()
    out["nnz_M"] = M._nnz()
    Kd_idx = K.indices()
    diag_mask = Kd_idx[0] == Kd_idx[1]
    dK = torch.zeros(n, dtype=torch.float64)
    dK[Kd_idx[0][diag_mask]] = K.values()[diag_mask]
    Md_idx = M.indices()
    dmask = Md_idx[0] == Md_idx[1]
    dM = torch.zeros(n, dtype=torch.float64)
    dM[Md_idx[0][dmask]] = M.values()[dmask]
    out["diag_K"] = dK.numpy()
    out["diag_M"] = dM.numpy()
    out["fro_K"] = float(torch.sqrt((K.values() ** 2).sum()))
    out["fro_M"] = float(torch.sqrt((M.values() ** 2).sum()))
    out["sum_M"] = float(M.values().sum())
    rowcnt = torch.bincount(Kd_idx[0], minlength=n)
    out["rowcnt_K"] = rowcnt.numpy().astype(np.int32)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, 3, generator=g, dtype=torch.float64)
    out["x_probe"] = x.numpy()
    out["Kx"] = torch.sparse.mm(K, x).numpy()
    out["Mx"] = torch.sparse.mm(M, x).numpy()
    if order == 2:
        out["o2_vertices"] = gt.tetmesh.vertices.numpy()
        out["o2_tets"] = gt.tetmesh.tets.numpy().astype(np.int32)

    gt.eigen_decomposition_arpack()
    out["arpack_all"] = np.concatenate([
        # rigid "eigenvalues" are U_hat_full Rayleigh quotients; recompute all k+6 for the record
        (gt.U_hat_full.T @ torch.sparse.mm(K, gt.U_hat_full)).diagonal().numpy()])
    out["eigenvalues"] = gt.eigenvalues.numpy()
    out["gt_freqs"] = gt.get_undamped_freqs().numpy()
    out["get_vals"] = gt.get_vals().numpy()
    U = gt.U_hat
    out["U_MU_diag"] = (U.T @ torch.sparse.mm(M, U)).diagonal().numpy()
    # K_lambda / K_mu quadratic forms (K is linear in the Lame parameters):
    # fit from two FixedLinear materials sharing the mesh.
    lam = MAT[1] * MAT[2] / ((1 + MAT[2]) * (1 - 2 * MAT[2]))
    mu = MAT[1] / (2 * (1 + MAT[2]))
    out["lame"] = np.asarray([lam, mu])
    if order == 1:
        # NB FixedLinear.forward is unreachable in the reference (no super().__init__()), so the
        # matrix-free stiff_func is probed on the trainable model below.
        out["U_hat_first4"] = U[:, :4].numpy()

    # --- task "material" / "mat_baseline": trainable model read-out + gradients (diff_model.py:371-388)
    if order == 1:
        for task in ("material", "mat_baseline"):
            torch.manual_seed(11)
            m = build_model(BOWL, mode_num=mode_num, order=order, mat=MAT, task=task)
            out[f"{task}_youngs_logits"] = m.material_model.youngs.probablity.detach().numpy().copy()
            out[f"{task}_poisson_logits"] = m.material_model.poisson.probablity.detach().numpy().copy()
            out[f"{task}_youngs_list"] = m.material_model.youngs_list.numpy()
            out[f"{task}_poisson_list"] = m.material_model.poisson_list.numpy()
            out[f"{task}_youngs"] = float(m.material_model.youngs())
            out[f"{task}_poisson"] = float(m.material_model.poisson())
            m.eigen_decomposition()
            out[f"{task}_eigenvalues"] = m.eigenvalues.numpy()
            with torch.no_grad():
                KU = torch.sparse.mm(m.stiff_matrix, m.U_hat)
                sf = m.stiff_func(m.U_hat.float()).double()
                out[f"{task}_stiff_func_relerr"] = float((sf - KU).norm() / KU.norm())
            f = m.get_undamped_freqs()
            out[f"{task}_freqs"] = f.detach().numpy()
            loss = f.sum()
            loss.backward()
            out[f"{task}_grad_youngs_logits"] = m.material_model.youngs.probablity.grad.numpy().copy()
            if task == "material":
                out[f"{task}_grad_poisson_logits"] = m.material_model.poisson.probablity.grad.numpy().copy()
            # full loop body: oscillator + MSE vs gt audio, gradient to logits (material_sync_train.py:139-167)
            forces = torch.zeros((1, 150))
            forces[0, 0] = 1
            osc = TraditionalDampedOscillator(forces, 1, mode_num, 8000, 32000, Material(MAT))
            gt_audio = osc(torch.from_numpy(out["gt_freqs"]).float())
            for p in m.material_model.parameters():
                p.grad = None
            f2 = m.get_undamped_freqs().float()
            sig = osc(f2 * 1.01)
            l2 = ((sig - gt_audio) ** 2).mean()
            l2.backward()
            out[f"{task}_loop_loss"] = float(l2)
            out[f"{task}_loop_grad_youngs_logits"] = m.material_model.youngs.probablity.grad.numpy().copy()
            if task == "material":
                out[f"{task}_loop_grad_poisson_logits"] = m.material_model.poisson.probablity.grad.numpy().copy()

    np.savez_compressed(os.path.join(HERE, f"g3_bowl_o{order}.npz"), **out)
    print(f"g3 bowl ord-{order} done in {time.time() - t0:.1f}s")


def geometry_backward():
    """G4: d(sum get_vals)/d(vertices) on a 4^3 cube, ord 1 and 2 (diff_model.py:390-399)."""
    verts, tets = kuhn_cube(4)
    out = {"verts": verts, "tets": tets, "mat": np.asarray(MAT)}
    for order in (1, 2):
        v = torch.from_numpy(verts).clone().requires_grad_(True)
        obj = DiffSoundObj(vertices=v, tets=torch.from_numpy(tets), mode_num=8, mat=MAT, order=order,
                           mat_model=FixedLinear, task="gt")
        obj.eigen_decomposition()
        vals = obj.get_vals()
        vals.sum().backward()
        out[f"o{order}_eigenvalues"] = obj.eigenvalues.numpy()
        out[f"o{order}_vals"] = vals.detach().numpy()
        out[f"o{order}_grad_vertices"] = v.grad.numpy()
        out[f"o{order}_U_hat"] = obj.U_hat.numpy()
    np.savez_compressed(os.path.join(HERE, "g4_cube4_geometry.npz"), **out)
    print("g4 done")


def oscillator():
    """G5: src/ddsp/oscillator.py:246-310 (Traditional) and :49-141 (Damped)."""
    out = {"mat": np.asarray(MAT)}
    g = torch.Generator().manual_seed(3)
    freqs = torch.sort(torch.rand(32, generator=g) * 9000 + 400)[0].reshape(32, 1)
    out["freqs"] = freqs.numpy()
    S, sr = 8000, 32000
    imp = torch.zeros((1, 150))
    imp[0, 0] = 1
    rnd = torch.randn((1, 150), generator=g)
    for name, force in (("impulse", imp), ("random", rnd)):
        osc = TraditionalDampedOscillator(force, 1, 32, S, sr, Material(MAT))
        f = freqs.clone().requires_grad_(True)
        sig = osc(f)
        out[f"trad_{name}_force"] = force.numpy()
        out[f"trad_{name}_signal"] = sig.detach().numpy()
        out[f"trad_{name}_damped_freq"] = osc.damped_freq[:, :, 0].detach().numpy()
        (sig ** 2).mean().backward()
        out[f"trad_{name}_grad_f"] = f.grad.numpy()
    # multi-audio DampedOscillator with learnable alpha/beta/amp
    torch.manual_seed(5)
    forces = torch.randn((3, 150), generator=g)
    dosc = DampedOscillator(forces, 3, 32, S, sr, [0.0, 1.0], Material(MAT))
    f = freqs.clone().requires_grad_(True)
    sig = dosc(f)
    out["damped_forces"] = forces.numpy()
    out["damped_alpha_params"] = dosc.alpha.params.detach().numpy()
    out["damped_beta_params"] = dosc.beta.params.detach().numpy()
    out["damped_alpha_values"] = dosc.alpha.values_list.numpy()
    out["damped_beta_values"] = dosc.beta.values_list.numpy()
    out["damped_amp_value"] = dosc.amp.value.detach().numpy()
    out["damped_alpha"] = dosc.alpha().detach().numpy()
    out["damped_beta"] = dosc.beta().detach().numpy()
    out["damped_amp"] = dosc.amp().detach().numpy()
    out["damped_signal"] = sig.detach().numpy()
    (sig ** 2).mean().backward()
    out["damped_grad_f"] = f.grad.numpy()
    out["damped_grad_alpha_params"] = dosc.alpha.params.grad.numpy()
    out["damped_grad_beta_params"] = dosc.beta.params.grad.numpy()
    out["damped_grad_amp_value"] = dosc.amp.value.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_oscillator.npz"), **out)
    print("g5 done")


def real_audio_front_half():
    """G7: src/ddsp/oscillator.py:178-243 (GTDampedOscillator.forward with and without
    the per-sample frequency offsets and the noise branch, damping()), :85-109 (DampedOscillator.early), :143-176
    (forward_curve), src/ddsp/filtered_noise.py:7-67 (FilteredNoise).  Inputs are stored or - the one large
    parameter, freq_nonlinear (A, m, S, bins) - regenerated from a stored NumPy seed."""
    A, m, S, sr = 2, 16, 4000, 32000
    f_range = [400.0, 1500.0, 4000.0, 9000.0]
    out = {"mat": np.asarray(MAT), "A": A, "m": m, "S": S, "sr": sr, "f_range": np.asarray(f_range)}
    g = torch.Generator().manual_seed(17)
    forces = torch.randn((A, 150), generator=g)
    out["forces"] = forces.numpy()
    torch.manual_seed(23)
    osc = GTDampedOscillator(forces, A, m, S, sr, f_range, Material(MAT))
    osc.noise.device = "cpu"  # (the reference's default device string is 'cuda')
    out["nl_seed"] = 99
    nl = np.random.default_rng(99).uniform(-4, 4, size=(A, m, S, len(f_range))).astype(np.float32)
    osc.freq_nonlinear.params.data.copy_(torch.from_numpy(nl))
    for k in ("freq_linear", "alpha", "beta"):
        out[f"gt_{k}_params"] = getattr(osc, k).params.detach().numpy().copy()
        out[f"gt_{k}_values"] = getattr(osc, k).values_list.numpy().copy()
    out["gt_amp_value"] = osc.amp.value.detach().numpy().copy()
    out["gt_noise_bank"] = osc.noise.coefficient_bank.detach().numpy().copy()
    out["gt_damping"] = osc.damping().detach().numpy()
    out["gt_freq_linear"] = osc.freq_linear().detach().numpy()
    nf = S // 64 + 1
    for tag, rate, nrate in (("lin", 0.0, 0.0), ("tv", 0.3, 0.0), ("tvn", 0.3, 1.0)):
        osc.zero_grad()
        torch.manual_seed(31)
        sig = osc(rate, nrate)
        out[f"gt_{tag}_signal"] = sig.detach().numpy()
        out[f"gt_{tag}_undamped_freq_mean"] = osc.undamped_freq.detach().mean(-1).numpy()
        (sig ** 2).mean().backward()
        for k in ("freq_linear", "alpha", "beta"):
            out[f"gt_{tag}_grad_{k}"] = getattr(osc, k).params.grad.numpy().copy()
        out[f"gt_{tag}_grad_amp"] = osc.amp.value.grad.numpy().copy()
        gnl = osc.freq_nonlinear.params.grad
        if gnl is not None and rate != 0.0:  # (A, m, S, bins): a strided sample and its sums over time
            out[f"gt_{tag}_grad_nl_sample"] = gnl[:, :, ::40, :].numpy().copy()
            out[f"gt_{tag}_grad_nl_tsum"] = gnl.double().sum(2).numpy()
        if nrate != 0.0:
            out[f"gt_{tag}_grad_noise_bank"] = osc.noise.coefficient_bank.grad.numpy().copy()
            torch.manual_seed(31)  # the draw FilteredNoise.forward made (filtered_noise.py:49-50)
            out[f"gt_{tag}_noise"] = (torch.rand(A, nf, 64, dtype=torch.float32) * 2 - 1).numpy()
    # FilteredNoise alone
    torch.manual_seed(41)
    fn = FilteredNoise(3, 8000, device="cpu")
    torch.manual_seed(43)
    y = fn()
    torch.manual_seed(43)
    out["fn_noise"] = (torch.rand(3, 8000 // 64 + 1, 64, dtype=torch.float32) * 2 - 1).numpy()
    out["fn_bank"] = fn.coefficient_bank.detach().numpy().copy()
    out["fn_signal"] = y.detach().numpy()
    (y ** 2).mean().backward()
    out["fn_grad_bank"] = fn.coefficient_bank.grad.numpy().copy()
    # DampedOscillator.early / forward_curve with a piecewise-linear damping table (material_real_train.py:151)
    from scipy import interpolate
    xs = np.array([270.0, 770.0, 1270.0, 5020.0, 9020.0])
    ys = np.array([3.0, 5.5, 4.0, 40.0, 90.0])
    table = interpolate.interp1d(xs, ys, fill_value="extrapolate")
    out["curve_x"], out["curve_y"] = xs, ys
    torch.manual_seed(47)
    dosc = DampedOscillator(forces, A, m, S, sr, [0.0, 1.0], Material(MAT))
    fl = torch.sort(torch.rand(m, generator=g) * 8000 + 300)[0].reshape(m, 1)
    out["curve_freqs"] = fl.numpy()
    for tag, fn_ in (("early", dosc.early), ("curve", dosc.forward_curve)):
        f = fl.clone().requires_grad_(True)
        sig = fn_(f, table)
        out[f"{tag}_signal"] = sig.detach().numpy()
        out[f"{tag}_damped_freq"] = dosc.damped_freq.detach().reshape(-1).numpy()
        (sig ** 2).mean().backward()
        out[f"{tag}_grad_f"] = f.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g7_real_audio.npz"), **out)
    print("g7 done", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.size > 1000})


def lobpcg_trajectory():
    """G6: reference lobpcg_func on the 2^3 cube ord-2 matrices (restatement check only)."""
    verts, tets = kuhn_cube(2)
    obj = DiffSoundObj(vertices=torch.from_numpy(verts), tets=torch.from_numpy(tets), mode_num=8, mat=MAT,
                       order=2, mat_model=FixedLinear, task="gt")
    obj.update_mass_matrix(MAT[0])
    obj.update_stiff_matrix()
    K = obj.stiff_matrix.float()
    M = obj.mass_matrix.float()
    torch.manual_seed(0)
    X0 = torch.randn(K.shape[0], 14)
    traj = []

    def tracker(w):
        traj.append(w.E.clone().numpy())

    torch.manual_seed(1)
    E, X = lobpcg_func(K, M, 14, X=X0.clone(), niter=50, largest=False, tracker=tracker)
    np.savez_compressed(os.path.join(HERE, "g6_lobpcg_ref.npz"), X0=X0.numpy(), E=E.numpy(),
                        traj=np.stack(traj), K=K.to_dense().numpy(), M=M.to_dense().numpy())
    print("g6 done")


def mesh_file_fixture():
    """G8 (round 5): one small Gmsh 2.2 binary file of the reference's data directory - data/mesh/shape/oloid.msh, 179 KB, DATA
    the reference ships (not source) - copied next to the fixtures, with what the REFERENCE's own loader
    (TetMesh.import_from_file, src/diffelastic/mesh.py:181-199: reader, float cast, remove_duplicate_vertices) makes of it.
    The product's byte-format reader / writer is checked against both (tests/test_host_logic.py)."""
    import shutil

    src = os.path.join(_ref_harness.REFERENCE_ROOT, "data/mesh/shape/oloid.msh")
    shutil.copyfile(src, os.path.join(HERE, "oloid.msh"))
    os.chmod(os.path.join(HERE, "oloid.msh"), 0o644)
    pts, tets = _ref_harness.read_gmsh22_binary(src)
    m = TetMesh().import_from_file(src)
    np.savez_compressed(os.path.join(HERE, "g8_oloid_import.npz"), raw_points=pts, raw_tets=tets.astype(np.int64),
                        vertices=m.vertices.numpy(), tets=m.tets.numpy().astype(np.int64), order=m.order)
    print("g8 done", m.vertices.shape, m.tets.shape)


def spectral_loss():
    """G9 (round 5): the reference's MSSLoss (src/ddsp/mss_loss.py:125-147 over SSSLoss :69-122 and weighted_l1_loss :50-62), types
    'l1_loss' (what material_sync_train.py:123-125 builds for its late epochs) and 'rmse_loss' (with and without a clipped band),
    run on two batches of decaying partials - value and gradient w.r.t. the predicted audio.  The module's third-party imports are
    absent from the image: _ref_harness.install_spectral() supplies torchaudio's Spectrogram as a restatement of torchaudio 2.0.2
    (what G9 pins is the reference's own arithmetic around it) and inert stand-ins for torchvision / geomloss."""
    _ref_harness.install_spectral()
    from src.ddsp.mss_loss import MSSLoss

    sr, n = 16000, 6000
    g = torch.Generator().manual_seed(91)
    t = torch.arange(n, dtype=torch.float64) / sr

    def partials(nm, seed_shift):
        f = 200.0 + 3800.0 * torch.rand((2, nm), generator=g, dtype=torch.float64)
        d = 4.0 + 60.0 * torch.rand((2, nm), generator=g, dtype=torch.float64)
        a = torch.rand((2, nm), generator=g, dtype=torch.float64) / nm
        return (a[..., None] * torch.exp(-d[..., None] * t) * torch.sin(2 * np.pi * f[..., None] * t)).sum(1).float()

    x_true = partials(12, 0)
    x_pred = (0.7 * x_true + 0.3 * partials(12, 1) + 1e-3 * torch.randn((2, n), generator=g)).contiguous()
    out = dict(x_pred=x_pred.numpy(), x_true=x_true.numpy(), sample_rate=sr)
    for tag, ffts, kind, scale in (("l1", [1024, 512, 256, 128, 64], "l1_loss", 1.0), ("l1_big", [2048, 1024], "l1_loss", 1.0),
                                   ("rmse", [1024, 512, 256, 128, 64], "rmse_loss", 1.0), ("rmse_half", [1024, 256], "rmse_loss", 0.5)):
        xp = x_pred.clone().requires_grad_(True)
        loss = MSSLoss(ffts, sr, type=kind)(xp, x_true, scale=scale)
        loss.backward()
        out[f"{tag}_n_ffts"], out[f"{tag}_scale"] = np.array(ffts), scale
        out[f"{tag}_loss"], out[f"{tag}_grad"] = loss.detach().double().numpy(), xp.grad.numpy()
        print("g9", tag, float(loss), float(xp.grad.abs().max()))
    np.savez_compressed(os.path.join(HERE, "g9_mss_loss.npz"), **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--skip-ord2-bowl", action="store_true")
    a = ap.parse_args()
    todo = a.only.split(",") if a.only else ["g1", "g2", "g3o1", "g3o2", "g4", "g5", "g6", "g7", "g8", "g9"]
    torch.set_num_threads(8)
    if "g8" in todo:
        mesh_file_fixture()
    if "g9" in todo:
        spectral_loss()
    if "g1" in todo:
        constants()
    if "g2" in todo:
        per_stage_cube()
    if "g5" in todo:
        oscillator()
    if "g6" in todo:
        lobpcg_trajectory()
    if "g7" in todo:
        real_audio_front_half()
    if "g4" in todo:
        geometry_backward()
    if "g3o1" in todo:
        bowl(1)
    if "g3o2" in todo and not a.skip_ord2_bowl:
        bowl(2, mode_num=32)
