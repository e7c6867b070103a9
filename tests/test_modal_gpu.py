"""End-to-end eigen-solve parity on the GPU: HIP path vs the reference's ARPACK eigenvalues
(golden fixtures) and vs the oracle on synthetic meshes.  pytest -m gpu."""
import numpy as np
import pytest
import torch

from oracle import fem, modal

pytestmark = pytest.mark.gpu
MAT = (2700.0, 5e10, 0.25, 6.0, 1e-7)
EIG_TOL = 1e-4  # stated fp32-solve tolerance on eigenvalues; measured ~1e-7


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def _solve(v, t, order, k, dev, **cfg):
    from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    sysd = TetSystem(v.to(dev), t.to(dev), order, MAT[0])
    lam, mu = fem.lame(MAT[1], MAT[2])
    ops = HipModalOps(sysd, lam, mu)
    res = ModalSolver(ops, SolverConfig(**cfg)).solve(k)
    return sysd, ops, res


@pytest.mark.parametrize("order,fixture", [(1, "g3_bowl_o1.npz"), (2, "g3_bowl_o2.npz")])
def test_bowl_eigenvalues_vs_reference(golden, dev, order, fixture):
    g = golden(fixture)
    m = golden("g0_bowl_mesh.npz")
    v, t = fem.to_high_order(torch.from_numpy(m["verts"]), torch.from_numpy(m["tets"]).long(), order)
    k = int(g["mode_num"])
    sysd, ops, res = _solve(v, t, order, k, dev)
    ev = res.eigenvalues.cpu().numpy()
    err = np.abs(ev - g["eigenvalues"]) / g["eigenvalues"]
    print("bowl ord", order, "iters", res.iterations, "max rel eig err", err.max())
    assert err.max() < EIG_TOL
    # read-out identities: lam*a + mu*b = lambda, u^T M u = 1
    lam, mu = ops.lame
    assert np.abs((lam * res.a_lambda + mu * res.b_mu).cpu().numpy() / ev - 1).max() < 1e-9
    assert np.abs(res.m_diag.cpu().numpy() - 1).max() < 1e-9
    # invariant-subspace check of the eigenvectors against M (mode order canonicalised by sorting)
    U = sysd.rows_to_external(res.vectors).double().cpu().numpy()
    K, M3 = sysd.to_scipy(lam, mu)
    R = K @ U - (M3 @ U) * ev[None, :]
    # backward-stable residual of the reference's convergence test (src/lobpcg/_lobpcg.py:318):
    # ||K u - lambda M u|| / (||u|| (||K|| + lambda ||M||)) with norms estimated on a random block
    G = np.random.default_rng(0).standard_normal((U.shape[0], 8))
    An = np.linalg.norm(K @ G) / np.linalg.norm(G)
    Bn = np.linalg.norm(M3 @ G) / np.linalg.norm(G)
    rerr = np.linalg.norm(R, axis=0) / (np.linalg.norm(U, axis=0) * (An + ev * Bn))
    assert rerr.max() < 1e-5
    # M-orthonormality of the returned modes
    assert np.abs(U.T @ (M3 @ U) - np.eye(U.shape[1])).max() < 1e-4


def test_cube_ord2_vs_oracle(dev):
    from diffsound_amd import meshgen

    v, t = meshgen.kuhn_box(6)
    v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), 2)
    d = fem.OracleDeform(v, t, 2)
    lam, mu = fem.lame(MAT[1], MAT[2])
    K = fem.assemble_stiffness(d, lam, mu)
    M3, _ = fem.assemble_mass(v, t, 2, MAT[0])
    ev_ref, _, _, _ = modal.eigsh_shift_invert(K, M3, 24)
    _, _, res = _solve(v, t, 2, 24, dev)
    err = np.abs(res.eigenvalues.cpu().numpy() - ev_ref) / ev_ref
    print("cube6 ord2 iters", res.iterations, "err", err.max())
    assert err.max() < EIG_TOL


def test_concurrent_hypothesis_lanes_match_sequential(dev):
    """ModalPipeline.run_batch with two hypotheses in flight (two HIP streams + host threads) returns what the
    sequential passes return: same eigenvalues, losses and (E, nu) gradients (hypotheses are independent)."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.lobpcg.modal_solver import SolverConfig
    from diffsound_amd.pipeline import ModalPipeline

    mat = (2700.0, 5e10, 0.25, 6.0, 1e-7)
    v, t = meshgen.kuhn_box(6)
    mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    pipe = ModalPipeline(mesh.vertices, mesh.tets, 2, 16, mat, solver_config=SolverConfig(block=24, lmax_cap=10.0))
    pipe.assemble()
    _, _, audio0 = pipe.run_pass(mat[1], mat[2], backward=False)
    pipe.set_target(audio0)
    hyps = [(4e10, 0.2), (7e10, 0.3), (5.5e10, 0.27), (9e10, 0.15), (3e10, 0.35)]
    seq = pipe.run_batch(hyps, lanes=1)
    par = pipe.run_batch(hyps, lanes=2)
    assert len(par) == len(hyps) and len(pipe._lanes) == 2
    for (rs, ress, as_), (rp, resp, ap) in zip(seq, par):
        assert np.abs(resp.eigenvalues.cpu().numpy() / ress.eigenvalues.cpu().numpy() - 1).max() < 1e-6
        assert abs(rp.loss / rs.loss - 1) < 1e-3
        assert abs(rp.grad_E / rs.grad_E - 1) < 1e-2 and abs(rp.grad_nu / rs.grad_nu - 1) < 1e-2
        assert float((ap - as_).norm() / as_.norm()) < 1e-3


def test_fp64_refinement_matches_arpack_to_1e9(dev):
    """The fp64 path of configs[4] at a size the oracle can solve: 8^3 ord-2 Kuhn box (n = 14 739), 32 modes,
    fp32 iterates + fp64 refinement to a backward error of 1e-10; eigenvalues against ARPACK's fp64 shift-invert
    values to 1e-9, fp64 M-orthonormal vectors, read-out identities to 1e-12."""
    from diffsound_amd import meshgen

    v, t = meshgen.kuhn_box(8)
    v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), 2)
    sysd, ops, res = _solve(v, t, 2, 32, dev, block=40, lmax_cap=10.0, refine_tol=1e-10)
    assert res.vectors.dtype == torch.float64 and 1 <= res.refine_iterations <= 40
    assert float(res.rerr.max()) < 1e-10
    lam, mu = ops.lame
    K, M3 = sysd.to_scipy(lam, mu)  # the HIP assembly itself (fp64), so ARPACK sees the same pencil
    ev_ref = modal.eigsh_shift_invert(K, M3, 32)[0]
    ev = res.eigenvalues.cpu().numpy()
    assert np.abs(ev / ev_ref - 1).max() < 1e-9
    U = sysd.rows_to_external(res.vectors).cpu().numpy()
    R = K @ U - (M3 @ U) * ev[None, :]
    G = np.random.default_rng(0).standard_normal((U.shape[0], 8))
    An, Bn = np.linalg.norm(K @ G) / np.linalg.norm(G), np.linalg.norm(M3 @ G) / np.linalg.norm(G)
    assert (np.linalg.norm(R, axis=0) / (np.linalg.norm(U, axis=0) * (An + ev * Bn))).max() < 2e-10
    assert np.abs(U.T @ (M3 @ U) - np.eye(32)).max() < 1e-10
    assert np.abs((lam * res.a_lambda + mu * res.b_mu).cpu().numpy() / ev - 1).max() < 1e-12
    # and against the oracle's own assembly (fp32 shape-function gradients: the 2e-6 assembly tolerance applies)
    d = fem.OracleDeform(v, t, 2)
    Ko = fem.assemble_stiffness(d, lam, mu)
    Mo, _ = fem.assemble_mass(v, t, 2, MAT[0])
    assert np.abs(ev / modal.eigsh_shift_invert(Ko, Mo, 32)[0] - 1).max() < 1e-5


@pytest.mark.parametrize("mesh,order,k,block,nested,storage", [(6, 2, 32, 40, 0.0, "bf16"), (6, 2, 32, 40, 1e-2, "bf16"),
                                                                (8, 1, 16, 24, 0.0, "bf16"), (6, 2, 32, 40, 1e-2, "fp32"),
                                                                (6, 2, 72, 84, 0.0, "bf16"), (8, 1, 16, 24, 0.0, "fp32")])
def test_native_iteration_driver_matches_python_loop(dev, mesh, order, k, block, nested, storage):
    """ds_lobpcg_iterate (the iteration as one native call, LAPACK from SciPy) against the Python loop it replaces:
    same kernels and dense steps, so the same iteration count and eigenvalues to the rounding of the two LAPACKs;
    two-level preconditioner (ord-2), nested start, one-level polynomial (ord-1), bf16 and fp32 preconditioner blocks,
    an 84-column block (the [X' P'] update then takes four launches instead of two)."""
    from diffsound_amd import meshgen
    from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(mesh)
    v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), order)
    sysd = TetSystem(v.to(dev), t.to(dev), order, MAT[0])
    lam, mu = fem.lame(MAT[1], MAT[2])
    # (fp32 preconditioner blocks: the corner-node level on its node blocks - the group-block Jacobi lives on the bf16 cycle and
    # an fp32 cycle would leave the native driver for the Python loop)
    ops = HipModalOps(sysd, lam, mu, coarse_group_jacobi=0 if storage == "fp32" else None)
    out = {}
    for native in (True, False):
        cfg = SolverConfig(block=block, lmax_cap=float({1: 4, 2: 10}[order]), tol=1e-5, nested_tol=nested, native=native,
                           precond_storage=storage)
        calls = []
        orig = ops.native_lobpcg
        ops.native_lobpcg = lambda *a, _o=orig, **kw: (calls.append(1), _o(*a, **kw))[1]
        try:
            out[native] = ModalSolver(ops, cfg).solve(k)
        finally:
            del ops.native_lobpcg
        assert bool(calls) == native  # the driver really ran (and only when asked)
    a, b_ = out[True], out[False]
    assert abs(a.iterations - b_.iterations) <= 1 and a.coarse_iterations == b_.coarse_iterations
    assert float(a.rerr.max()) < 1e-5 and float(b_.rerr.max()) < 1e-5
    assert float((a.eigenvalues / b_.eigenvalues - 1).abs().max()) < 1e-6
    assert len(a.history) == a.iterations + 1 and a.history[-1][1] < 1e-5


@pytest.mark.parametrize("native", [True, False])
def test_fresh_products_and_fused_residual_change_nothing_but_rounding(dev, native):
    """The three forms of the iteration's K X' / residual step - (a) K X' by the update of K [X P W] (round 3), (b) K X' by one
    fresh product (kx_fresh), (c) residual and norms in one walk of the unions (fused_residual, the default) - are the same
    mathematics: the same iteration counts (within one) and eigenvalues to the solver's accuracy; (b) and (c) form the SAME
    residual bit for bit, so they agree far more closely than that."""
    from diffsound_amd import meshgen
    from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(8)
    v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), 2)
    sysd = TetSystem(v.to(dev), t.to(dev), 2, MAT[0])
    lam, mu = fem.lame(MAT[1], MAT[2])
    res = {}
    for name, kw in (("recurrence", dict(kx_fresh=False, fused_residual=False)), ("fresh", dict(kx_fresh=True, fused_residual=False)),
                     ("fused", dict(kx_fresh=True, fused_residual=True))):
        ops = HipModalOps(sysd, lam, mu)  # fresh operators: no state carried from one form to the next
        cfg = SolverConfig(block=40, lmax_cap=10.0, tol=1e-5, nested_tol=1e-2, native=native, **kw)
        res[name] = ModalSolver(ops, cfg).solve(32)
        assert float(res[name].rerr.max()) < 1e-5
    for name in ("fresh", "fused"):
        assert abs(res[name].iterations - res["recurrence"].iterations) <= 1
        assert float((res[name].eigenvalues / res["recurrence"].eigenvalues - 1).abs().max()) < 1e-6
    # (the residual block is the same bit for bit - tests/test_hip_kernels.py - but its column norms are summed in another order, and
    # a pair whose backward error sits at the tolerance is locked an iteration earlier by one form than by the other: the
    # trajectories part in the last digits - measured round 6: 1.049e-5 against 1.056e-5 at the last test but one)
    assert abs(res["fused"].iterations - res["fresh"].iterations) <= 1
    assert float((res["fused"].eigenvalues / res["fresh"].eigenvalues - 1).abs().max()) < 1e-6


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("mesh,order,k,block,nested", [(8, 2, 32, 40, 1e-2), (8, 2, 64, 80, 3e-3), (10, 1, 20, 28, 0.0)])
def test_rayleigh_ritz_on_the_raw_basis_changes_nothing_but_rounding(dev, native, mesh, order, k, block, nested):
    """Round 5: SolverConfig.raw_rr - K W and M W of the raw preconditioned residuals in one walk, ONE Gram launch, the
    orthonormalisation of W folded into the small dense algebra and ONE update from the raw basis - is the same mathematics
    as the explicit sequence (M W, Gram, update of W, K W, Gram, update): the same iteration counts (within one), eigenvalues
    to the solver's accuracy, and against ARPACK to the stated tolerance; per iteration it must really issue fewer products."""
    from diffsound_amd import meshgen
    from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(mesh)
    v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), order)
    sysd = TetSystem(v.to(dev), t.to(dev), order, MAT[0])
    lam, mu = fem.lame(MAT[1], MAT[2])
    res, counts = {}, {}
    for raw in (False, True):
        ops = HipModalOps(sysd, lam, mu)
        cfg = SolverConfig(block=block, lmax_cap=float({1: 4, 2: 10}[order]), tol=1e-5, nested_tol=nested, native=native, raw_rr=raw)
        res[raw] = ModalSolver(ops, cfg).solve(k)
        counts[raw] = dict(ops.counts)
        assert float(res[raw].rerr.max()) < 1e-5
    assert abs(res[True].iterations - res[False].iterations) <= 1
    assert float((res[True].eigenvalues / res[False].eigenvalues - 1).abs().max()) < 1e-6
    if not native:  # (the Python loop counts its launches: two Gram products and two updates per iteration became one each)
        assert counts[True]["gram"] < counts[False]["gram"] and counts[True]["mix"] < counts[False]["mix"]
    d = fem.OracleDeform(v, t, order)
    K = fem.assemble_stiffness(d, lam, mu)
    M3, _ = fem.assemble_mass(v, t, order, MAT[0])
    ev = modal.eigsh_shift_invert(K, M3, k)[0]
    assert float(np.abs(res[True].eigenvalues.cpu().numpy() / ev - 1).max()) < EIG_TOL


def test_norm_probe_and_power_block_are_kept_per_geometry(dev):
    """What a pass re-uses from the previous hypothesis on the same geometry (round 5): the random probe block of the operator-norm
    estimates with ||M G0|| / ||G0|| - same numbers as a fresh computation, bit for bit - and the power iteration's block; new
    coordinates start a new generation, and the estimates are recomputed."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig, SolverState
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(6)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    sysd = TetSystem(tm.vertices, tm.tets, 2, MAT[0])
    lam, mu = fem.lame(MAT[1], MAT[2])
    ops = HipModalOps(sysd, lam, mu)
    cfg = SolverConfig(block=24, tol=1e-5)

    def solve():
        st = SolverState({}, {}, {})
        res = ModalSolver(ops, cfg).solve(16, state=st)
        return st.fvars["A_norm"], st.fvars["B_norm"], res.eigenvalues.clone()

    a0, b0, e0 = solve()
    key0, G0 = ops._norm_probe[0], ops._norm_probe[1]
    # ||K G0|| formed from the kept geometry-only products lam K_lambda G0 + mu K_mu G0 (fp64 values) = the fp32 product's norm
    G1 = torch.empty_like(G0)
    ops.apply_K(G0, G1)
    assert abs(float(torch.linalg.vector_norm(G1.double()) / torch.linalg.vector_norm(G0.double())) / a0 - 1) < 1e-5
    ops.apply_M(G0, G1)
    assert abs(float(torch.linalg.vector_norm(G1.double()) / torch.linalg.vector_norm(G0.double())) / b0 - 1) < 1e-5
    a1, b1, e1 = solve()  # same material, same geometry: the kept block, identical estimates (the interval of the
    # preconditioner now comes from the warm power block, so the iterates differ in rounding)
    assert ops._norm_probe[1] is G0 and (a1, b1) == (a0, b0) and torch.allclose(e0, e1, rtol=1e-6)
    del ops._norm_probe
    a2, b2, _ = solve()   # recomputed from scratch: the same numbers
    assert (a2, b2) == (a0, b0)
    lam2, mu2 = fem.lame(2 * MAT[1], 0.3)
    ops.set_material(lam2, mu2)
    sysd.assemble()       # numeric assembly on the same coordinates: same generation
    a3, b3, _ = solve()
    assert ops._norm_probe[0] == key0 and b3 == b0 and a3 != a0
    warm_before = list(ops.warm_stats)[0]
    solve()
    assert ops.warm_stats[0] > warm_before  # (same geometry: the intervals' ends come from the kept power block)
    sysd.assemble(tm.vertices * 1.1)  # (coordinates in the caller's numbering)
    ops.set_material(lam2, mu2)
    warm_before = ops.warm_stats[0]
    a4, b4, _ = solve()
    assert ops.warm_stats[0] == warm_before  # (new coordinates: the power iteration starts over, no warm estimate)
    assert ops._norm_probe[0] != key0 and abs(b4 / b0 - 1.1 ** 3) < 1e-3  # (mass entries scale with the volume)


def test_rigid_basis_follows_the_geometry(dev):
    """An operator object that outlives a geometry update (DiffSoundObj.update_mass_matrix + update_stiff_matrix in a shape
    loop) re-forms its rigid-body basis: rotations are fields of the coordinates, and the basis is M-orthonormal in the NEW mass
    matrix.  Checked on a stretched mesh: K Y = 0 to rounding, Y^T M Y = I, and the solve finds the same elastic spectrum as
    an operator object built on the stretched mesh from scratch."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(5)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    lam, mu = fem.lame(MAT[1], MAT[2])
    sysd = TetSystem(tm.vertices, tm.tets, 2, MAT[0])
    ops = HipModalOps(sysd, lam, mu)
    stretched = tm.vertices * torch.tensor([1.4, 1.0, 0.8], device=dev) + torch.tensor([0.3, -0.2, 0.1], device=dev)
    sysd.assemble(stretched)
    ops.set_material(lam, mu)

    def defect(o):
        Y = o.rigid[:, :6].contiguous()
        Y8 = o.rigid.contiguous()
        KY, MY = torch.empty_like(Y8), torch.empty_like(Y8)
        o.apply_K(Y8, KY)
        o.apply_M(Y8, MY)
        G = (Y.double().T @ MY[:, :6].double())
        probe = torch.randn((o.n, 8), device=dev)
        Kp = torch.empty_like(probe)
        o.apply_K(probe, Kp)
        knorm = float(torch.linalg.vector_norm(Kp.double()) / torch.linalg.vector_norm(probe.double()))
        return float(torch.linalg.vector_norm(KY[:, :6].double(), dim=0).max()) / knorm, float((G - torch.eye(6, device=dev, dtype=torch.float64)).abs().max())

    kdef, gdef = defect(ops)
    assert kdef < 1e-4 and gdef < 1e-5, (kdef, gdef)  # (measured 9e-6: fp32 rounding of K Y; the previous geometry's basis gives 0.18)
    fresh = HipModalOps(TetSystem(stretched, tm.tets, 2, MAT[0]), lam, mu)
    cfg = SolverConfig(block=24, tol=1e-6)
    e_kept = ModalSolver(ops, cfg).solve(16).eigenvalues
    e_fresh = ModalSolver(fresh, cfg).solve(16).eigenvalues
    assert float(((e_kept - e_fresh).abs() / e_fresh).max()) < 1e-5
    assert float(e_kept[0]) > 1e6  # (no rigid mode leaked into the elastic spectrum)


def test_in_place_vertex_update_without_reordering_is_seen(dev):
    """with ``reorder=False`` and an fp32 contiguous input the system used to keep the CALLER's storage as its snapshot
    of the coordinates; a caller that then moved its vertices in place and called ``assemble(vertices)`` compared the tensor with
    itself - no new generation, the corner-node level re-assembled on its old copy, the rigid basis stale.  The snapshot is a
    private copy now: the in-place move is seen on both levels and the solve agrees with a system built on the moved mesh."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(5)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    lam, mu = fem.lame(MAT[1], MAT[2])
    verts = tm.vertices.float().contiguous().clone()  # the caller's own array, moved in place below
    sysd = TetSystem(verts, tm.tets, 2, MAT[0], reorder=False)
    assert sysd.vertices.data_ptr() != verts.data_ptr()
    ops = HipModalOps(sysd, lam, mu)
    gen0 = getattr(sysd, "geometry_generation", 0)
    cgen0 = getattr(sysd._coarse["sys"], "geometry_generation", 0)
    verts.mul_(torch.tensor([1.4, 1.0, 0.8], device=dev)).add_(torch.tensor([0.3, -0.2, 0.1], device=dev))
    sysd.assemble(verts)
    assert sysd.vertices.data_ptr() != verts.data_ptr()
    assert getattr(sysd, "geometry_generation", 0) == gen0 + 1
    assert getattr(sysd._coarse["sys"], "geometry_generation", 0) == cgen0 + 1
    assert torch.equal(sysd._coarse["sys"].vertices, verts[sysd._coarse["corners"]])
    ops.set_material(lam, mu)
    fresh = HipModalOps(TetSystem(verts.clone(), tm.tets, 2, MAT[0], reorder=False), lam, mu)
    assert torch.equal(sysd.klam, fresh.sys.klam) and torch.equal(sysd._coarse["sys"].klam, fresh.sys._coarse["sys"].klam)
    cfg = SolverConfig(block=24, tol=1e-6)
    e_kept = ModalSolver(ops, cfg).solve(16).eigenvalues
    e_fresh = ModalSolver(fresh, cfg).solve(16).eigenvalues
    assert float(((e_kept - e_fresh).abs() / e_fresh).max()) < 1e-5
    assert float(e_kept[0]) > 1e6
    # the same coordinates again: no new generation (DiffSoundObj.eigen_decomposition hands them over on every call) - and no new
    # assembly either (round 6: K_lambda, K_mu, M_s depend on the geometry alone and are in place)
    skipped = getattr(sysd, "assemblies_skipped", 0)
    kept = sysd.klam.clone()
    sysd.klam.zero_()  # (would be rewritten by an assembly)
    sysd.assemble(verts)
    assert getattr(sysd, "geometry_generation", 0) == gen0 + 1 and sysd.assemblies_skipped == skipped + 1
    assert float(sysd.klam.abs().max()) == 0.0
    sysd.klam.copy_(kept)
    sysd.assemble()    # without coordinates (a pass of the pipeline): always assembles
    assert torch.equal(sysd.klam, kept)


def test_host_wait_mode_changes_nothing_but_the_waiting(dev):
    """ds_host_wait_mode: the native solve's waits for its stream as a poll followed by a sleep on a blocking event (what the lane
    pool of the pipeline selects) against hipStreamSynchronize - the same results bit for bit; other modes are refused."""
    from diffsound_amd import _hip

    L = _hip.lib()
    from diffsound_amd import meshgen

    vv, tt = meshgen.kuhn_box(5)
    vv, tt = torch.from_numpy(vv), torch.from_numpy(tt).long()
    try:
        assert L.ds_host_wait_mode(0) == 0
        _, _, r0 = _solve(vv, tt, 1, 12, dev, block=16, tol=1e-5)
        assert L.ds_host_wait_mode(1) == 0
        _, _, r1 = _solve(vv, tt, 1, 12, dev, block=16, tol=1e-5)
        assert torch.equal(r0.eigenvalues, r1.eigenvalues) and torch.equal(r0.vectors, r1.vectors) and r0.iterations == r1.iterations
        assert L.ds_host_wait_mode(2) != 0 and L.ds_last_error()
    finally:
        L.ds_host_wait_mode(0)


@pytest.mark.parametrize("mesh,order,k,block,nested", [(6, 2, 16, 24, 3e-3), (8, 1, 12, 16, 0.0), (5, 2, 8, 16, 0.0)])
def test_start_block_in_coefficients_changes_nothing_but_rounding(dev, mesh, order, k, block, nested):
    """The start block's projection, orthonormalisation and first Ritz step from one [K X0 | M X0] walk, one Gram launch and one
    update (SolverConfig.raw_start, round 5) against the explicit sequence: the same eigenvalues to the solve's accuracy, the same
    iteration counts, and the start was taken (not handed to the explicit route) on these well-conditioned blocks."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.lobpcg.modal_solver import ModalSolver

    v, t = meshgen.kuhn_box(mesh)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    out = {}
    for raw in (True, False):
        _, ops, res = _solve(tm.vertices, tm.tets, order, k, dev, block=block, tol=1e-6, nested_tol=nested, raw_start=raw)
        out[raw] = res
        assert (getattr(ops, "raw_start_stats", [0, 0])[0] > 0) == raw  # (the counters of THIS operator object, round 6)
    a, b = out[True], out[False]
    assert float(((a.eigenvalues - b.eigenvalues).abs() / b.eigenvalues).max()) < 2e-6
    assert abs(a.iterations - b.iterations) <= 1
    ref = modal.reference_eigs(ops, k) if hasattr(modal, "reference_eigs") else None
    assert ref is None or float(((a.eigenvalues.cpu() - ref).abs() / ref).max()) < EIG_TOL


def test_group_block_jacobi_pieces_and_polynomial(dev):
    """The group-block Jacobi of the corner-node level's polynomial (round 6; ds_group_inverse / ds_group_pack_kc / ds_group_apply16,
    ds_level_t.tgrp): T_g against torch's inverse of the 24 x 24 diagonal blocks gathered from the BSR values; T_g X against
    torch.bmm; and the polynomial p(T_g K) T_g R of the native bf16 driver - the term kernel on the DENSE blocks of T_g K, an identity
    for dinv, the right-hand side through T_g first - against the same polynomial in torch on the level's fp32 product, to the
    rounding of bf16 iterates, every column, deterministic.  On a mesh whose corner level ends in a group of fewer than 8 nodes."""
    from diffsound_amd import meshgen
    from diffsound_amd.lobpcg.modal_solver import ChebyshevBlockJacobi
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(10)
    v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), 2)
    sysd = TetSystem(v.to(dev), t.to(dev), 2, MAT[0])
    lam, mu = fem.lame(MAT[1], MAT[2])
    ops = HipModalOps(sysd, lam, mu)
    co = ops.coarse
    assert co.group_jacobi == 8 and ops.group_jacobi == 0 and co.nv % 8 != 0
    s, G = co.sys, 8
    ng = (s.nv + G - 1) // G
    rows = torch.repeat_interleave(torch.arange(s.nv, device=dev), (s.rowptr[1:] - s.rowptr[:-1]).long())
    cols = s.colidx.long()
    sel = (rows // G) == (cols // G)
    Kgg = torch.zeros((ng, 24, 24), dtype=torch.float64, device=dev)
    blk = co.k32[sel].double().reshape(-1, 3, 3)
    g, a, b = rows[sel] // G, rows[sel] % G, cols[sel] % G
    for i in range(3):
        for j in range(3):
            Kgg[g, 3 * a + i, 3 * b + j] = blk[:, i, j]
    idx = torch.arange(3 * (s.nv - G * (ng - 1)), 24, device=dev)
    Kgg[-1, idx, idx] = 1.0
    Tref = torch.linalg.inv(Kgg)
    err = (co.tgrp.double() - Tref).abs().amax(dim=(1, 2)) / Tref.abs().amax(dim=(1, 2))
    assert float(err.max()) < 1e-6  # (fp64 Gauss-Jordan, stored in fp32)
    assert torch.equal(co.tgrp, co.tgrp.transpose(1, 2))
    X = torch.randn((co.n, 12), device=dev)
    Xp = torch.cat([X, X.new_zeros((ng * 24 - co.n, 12))], 0)
    ref = torch.bmm(co.tgrp, Xp.reshape(ng, 24, 12)).reshape(-1, 12)[:co.n]
    assert float((co.group_T(X) - ref).abs().max() / ref.abs().max()) < 1e-6
    pre = ChebyshevBlockJacobi(co, 14, 150.0, 20, 0, 1.2, cap=4.0)
    assert pre.group == 8 and 1.5 < pre.lmax <= 4.0
    for ncols in (80, 40):
        R = torch.randn((co.n, ncols), device=dev) * 1e9
        Wn = torch.full_like(R, float("nan"))
        assert co.chebyshev_apply16(pre, R.clone(), Wn)  # the native bf16 driver
        Wp = torch.empty_like(R)
        pre.apply(R.clone(), Wp)  # torch, fp32
        cols_err = (Wn - Wp).norm(dim=0) / Wp.norm(dim=0)
        assert bool(torch.isfinite(Wn).all()) and float(cols_err.max()) < 3e-2, float(cols_err.max())
        Wn2 = torch.empty_like(R)
        co.chebyshev_apply16(pre, R.clone(), Wn2)
        assert torch.equal(Wn, Wn2)
    # a group whose diagonal block is not positive definite (here: one node's own block negated) falls back to the inverses of its
    # nodes' 3 x 3 blocks; the other groups are untouched
    from diffsound_amd import _hip
    kbad = co.k32.clone()
    node = 8 * 5 + 2
    j = int(s.rowptr[node]) + int((s.colidx[int(s.rowptr[node]):int(s.rowptr[node + 1])] == node).nonzero()[0])
    kbad[j] = -kbad[j]
    Tb = torch.empty_like(co.tgrp)
    _hip.check(_hip.lib().ds_group_inverse(_hip.ptr(s.rowptr), _hip.ptr(s.colidx), _hip.ptr(kbad), s.nv, 8, _hip.ptr(Tb),
                                           _hip.stream_ptr()), "ds_group_inverse")
    keep = torch.ones(ng, dtype=torch.bool, device=dev)
    keep[5] = False
    assert torch.equal(Tb[keep], co.tgrp[keep])
    want = torch.zeros((24, 24), dtype=torch.float64, device=dev)
    for a_ in range(8):
        nd = 8 * 5 + a_
        jj = int(s.rowptr[nd]) + int((s.colidx[int(s.rowptr[nd]):int(s.rowptr[nd + 1])] == nd).nonzero()[0])
        want[3 * a_:3 * a_ + 3, 3 * a_:3 * a_ + 3] = torch.linalg.inv(kbad[jj].double().reshape(3, 3))
    assert float((Tb[5].double() - want).abs().max() / want.abs().max()) < 1e-6
    # a new material: T_g and the blocks of T_g K follow (the polynomial of the OLD material on the new blocks would not match)
    co.set_material(2.0 * lam, 0.7 * mu)
    pre2 = ChebyshevBlockJacobi(co, 14, 150.0, 20, 0, 1.2, cap=4.0)
    R = torch.randn((co.n, 80), device=dev) * 1e9
    Wn, Wp = torch.empty_like(R), torch.empty_like(R)
    assert co.chebyshev_apply16(pre2, R.clone(), Wn)
    pre2.apply(R.clone(), Wp)
    assert float(((Wn - Wp).norm(dim=0) / Wp.norm(dim=0)).max()) < 3e-2


def test_group_block_jacobi_keeps_the_iteration_counts_with_a_shorter_polynomial(dev):
    """The corner-node level's polynomial on the group-block Jacobi, Chebyshev(14, ratio 150) in T_g K, against the node blocks'
    Chebyshev(22, ratio 350) at the benchmark's settings: the same eigenpairs to the solve's accuracy, no more iterations on either
    level (one of slack), through the native loop and through the Python loop (whose polynomial is the torch form on fp32)."""
    from diffsound_amd import meshgen
    from diffsound_amd.lobpcg.modal_solver import ModalSolver
    from diffsound_amd.modal_ops import HipModalOps, TetSystem
    import bench

    v, t = meshgen.kuhn_box(16)
    v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), 2)
    sysd = TetSystem(v.to(dev), t.to(dev), 2, MAT[0])
    lam, mu = fem.lame(MAT[1], MAT[2])
    res = {}
    for gj in (0, 8):
        ops = HipModalOps(sysd, lam, mu, coarse_group_jacobi=gj)
        assert ops.coarse.group_jacobi == gj
        for native in (True, False):
            cfg = bench.solver_config()
            cfg.native = native
            assert (cfg.group_degree, cfg.group_ratio, cfg.coarse_degree, cfg.coarse_ratio) == (14, 150.0, 22, 350.0)
            res[gj, native] = ModalSolver(ops, cfg).solve(64)
            print("group", gj, "native", native, "corner", res[gj, native].coarse_iterations, "fine", res[gj, native].iterations)
    ref = res[0, True]
    for key, r in res.items():
        assert float(r.rerr.max()) < 1e-5
        assert float(((r.eigenvalues - ref.eigenvalues).abs() / ref.eigenvalues).max()) < 1e-5
        assert r.iterations <= ref.iterations + 1 and r.coarse_iterations <= ref.coarse_iterations + 1, key


def test_one_level_polynomial_on_the_group_blocks(dev):
    """HipModalOps(one_level_group_jacobi=8): the ONE-level polynomial of an ord-1 mesh's operator object on the group-block Jacobi
    (off by default - the shape loop's fresh objects pay more for the blocks than the shorter polynomial saves,
    profiles/r06_geom_one_level_group.txt): the same eigenpairs as on the node blocks, no more iterations at two thirds of the terms."""
    from diffsound_amd import meshgen
    from diffsound_amd.lobpcg.modal_solver import ModalSolver, tuned_config
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(20)
    sysd = TetSystem(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev), 1, MAT[0])
    lam, mu = fem.lame(MAT[1], MAT[2])
    res = {}
    for gj in (0, 8):
        ops = HipModalOps(sysd, lam, mu, one_level_group_jacobi=gj)
        assert ops.group_jacobi == gj and ops.coarse is None
        cfg = tuned_config(1, tol=1e-5)
        assert (cfg.cheb_degree, cfg.cheb_group_degree) == (24, 16)
        s = ModalSolver(ops, cfg)
        assert s.precond.degree == (16 if gj else 24) and s.precond.group == gj
        res[gj] = s.solve(32)
    a, b = res[0], res[8]
    assert float(b.rerr.max()) < 1e-5 and float(((a.eigenvalues - b.eigenvalues).abs() / a.eigenvalues).max()) < 1e-5
    assert b.iterations <= a.iterations + 1


def test_swept_start_block_is_not_locked_before_its_ritz_values_have_settled(dev):
    """SolverConfig.nested_ritz_tol (round 6).  The corner-node phase of a nested start stops on a backward error of 3e-3 relative to
    ||K|| + lambda ||M|| - a test a SMOOTH block passes whatever its Rayleigh quotients are.  A start block that went through the
    preconditioner is smooth: with 32 modes in a block of 40 its first 16 columns were locked at the first test with Ritz values
    2 x off, the phase ran to its iteration cap and the fine level took twice the iterations of a plain random start
    (profiles/r06_start_sweeps.txt).  With the settled test the swept start costs no more fine-level iterations than the plain one
    (one of slack), its corner phase ends before the cap, and the native loop and the Python loop agree on both counts."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    import bench

    v, t = meshgen.kuhn_box(26)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    run = {}
    for name, over in (("plain", dict(start_sweeps=0)), ("swept", dict(start_sweeps=2)), ("swept, python loop", dict(start_sweeps=2, native=False)),
                       ("swept, backward error alone", dict(start_sweeps=2, nested_ritz_tol=0.0))):
        cfg = bench.solver_config(block=40)
        for k_, v_ in over.items():
            setattr(cfg, k_, v_)
        assert cfg.nested_ritz_tol == (0.0 if "alone" in name else 0.2)
        _, _, run[name] = _solve(tm.vertices, tm.tets, 2, 32, dev, **cfg.__dict__)
        print(name, "corner", run[name].coarse_iterations, "fine", run[name].iterations)
    plain, swept, py = run["plain"], run["swept"], run["swept, python loop"]
    assert float(swept.rerr.max()) < 1e-5 and float(plain.rerr.max()) < 1e-5
    assert swept.coarse_iterations < 8 and swept.iterations <= plain.iterations + 1
    assert abs(py.coarse_iterations - swept.coarse_iterations) <= 1 and abs(py.iterations - swept.iterations) <= 1
    assert float(((swept.eigenvalues - plain.eigenvalues).abs() / plain.eigenvalues).max()) < 1e-5
    # (what the settled test repairs - not asserted as a failure, it is a property of this mesh: the old test's counts are printed)
    assert run["swept, backward error alone"].iterations >= swept.iterations


def test_native_solve_on_the_lapack_inside_libtorch(dev):
    """The native iteration with the MKL entry points of libtorch_cpu.so (dsyevd / dgemm, no stages) instead of SciPy's capsules -
    the fallback that keeps the product independent of SciPy's private table: the same eigenvalues to the solve's accuracy, the
    same iteration counts within one."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh

    v, t = meshgen.kuhn_box(6)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    saved = (_hip.lapack_table(), _hip.lapack_source())
    out = {}
    try:
        for src in ("scipy", "torch"):
            _hip._set_lapack(_hip.lapack_table(src), src)
            assert _hip.lapack_source() == src
            _, _, out[src] = _solve(tm.vertices, tm.tets, 2, 16, dev, block=24, tol=1e-6, nested_tol=3e-3, start_sweeps=2)
    finally:
        _hip._set_lapack(*saved)
    a, b = out["scipy"], out["torch"]
    assert float(((a.eigenvalues - b.eigenvalues).abs() / b.eigenvalues).max()) < 2e-6
    assert abs(a.iterations - b.iterations) <= 1 and float(b.rerr.max()) < 1e-6
