"""Kernel-level parity: every C-ABI entry point against the CPU oracle on the same seeded inputs.
Run on the GPU box with  pytest -m gpu."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import fem
from oracle import oscillator as oosc
from oracle.ops_cpu import CpuModalOps

pytestmark = pytest.mark.gpu

MAT = (2700.0, 5e10, 0.25, 6.0, 1e-7)


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def _mesh(name, order):
    from diffsound_amd import meshgen

    if name == "bowl":
        m = np.load("tests/golden/g0_bowl_mesh.npz")
        v, t = torch.from_numpy(m["verts"]), torch.from_numpy(m["tets"]).long()
    else:
        v, t = meshgen.kuhn_box(int(name[4:]))
        v, t = torch.from_numpy(v), torch.from_numpy(t).long()
    return fem.to_high_order(v, t, order)


@pytest.fixture(scope="module", params=[("cube3", 1), ("cube3", 2), ("bowl", 1), ("cube6", 2)])
def case(request, dev):
    from diffsound_amd.modal_ops import TetSystem, HipModalOps

    name, order = request.param
    v, t = _mesh(name, order)
    d = fem.OracleDeform(v, t, order)
    Kl = fem.assemble_stiffness(d, 1.0, 0.0)
    Km = fem.assemble_stiffness(d, 0.0, 1.0)
    M3, Ms = fem.assemble_mass(v, t, order, MAT[0])
    lam, mu = fem.lame(MAT[1], MAT[2])
    sysd = TetSystem(v.to(dev), t.to(dev), order, MAT[0], reorder=False)  # kernel parity in the caller's numbering
    # (kernel parity against the oracle's operators: the corner-node level on the 3 x 3 node blocks, as the oracle's polynomial is;
    # the group-block Jacobi has its own tests - test_group_block_jacobi_*)
    hops = HipModalOps(sysd, lam, mu, coarse_group_jacobi=0)
    cops = CpuModalOps(Kl, Km, M3, v.numpy(), lam, mu, tets=t.numpy() if order == 2 else None)
    return dict(v=v, t=t, order=order, Kl=Kl, Km=Km, M3=M3, Ms=Ms, sys=sysd, hops=hops, cops=cops, lam=lam, mu=mu)


def test_pattern_and_assembly(case):
    s = case["sys"]
    Kl, Km, Ms = s.to_scipy()
    # pattern == union of the element cliques (same nnz as the reference's coalesced K)
    t = case["t"].numpy()
    N = t.shape[1]
    pairs = np.unique(np.stack([np.repeat(t, N, axis=1).reshape(-1), np.tile(t, (1, N)).reshape(-1)], 1), axis=0)
    ref_pat = sp.csr_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(s.nv, s.nv))
    ref_pat.sort_indices()
    assert s.nnzb == ref_pat.nnz
    assert np.array_equal(s.rowptr.cpu().numpy(), ref_pat.indptr)
    assert np.array_equal(s.colidx.cpu().numpy(), ref_pat.indices)
    # values: relative to the largest entry; 2e-6 absorbs the reference's fp32 shape-function gradients
    x = np.random.default_rng(0).standard_normal((s.n, 4))
    assert rel(Kl @ x, case["Kl"] @ x) < 2e-6
    assert rel(Km @ x, case["Km"] @ x) < 2e-6
    assert rel(sp.kron(Ms, sp.identity(3)) @ x, case["M3"] @ x) < 1e-12
    # symmetry and rigid-body null space
    K = case["lam"] * Kl + case["mu"] * Km
    assert abs(K - K.T).max() / abs(K).max() < 1e-12


def test_morton_reordering_is_transparent(case, dev):
    """Internal Morton renumbering: same matrices in the caller's numbering, same products."""
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    s0 = case["sys"]
    s1 = TetSystem(case["v"].to(dev), case["t"].to(dev), case["order"], MAT[0], reorder=True)
    assert sorted(s1.perm.cpu().tolist()) == list(range(s0.nv))
    K0, M0 = s0.to_scipy(case["lam"], case["mu"])
    K1, M1 = s1.to_scipy(case["lam"], case["mu"])
    assert abs(K0 - K1).max() / abs(K0).max() < 1e-14 and abs(M0 - M1).max() / abs(M0).max() < 1e-14
    h1 = HipModalOps(s1, case["lam"], case["mu"])
    g = torch.Generator().manual_seed(3)
    X = torch.randn((s0.n, 24), generator=g).to(dev)
    Y0 = torch.empty_like(X)
    case["hops"].apply_K(X, Y0)
    Xi = s1.rows_to_internal(X).contiguous()
    Yi = torch.empty_like(Xi)
    h1.apply_K(Xi, Yi)
    assert rel(s1.rows_to_external(Yi).cpu().numpy(), Y0.cpu().numpy()) < 5e-6
    assert torch.equal(s1.rows_to_external(s1.rows_to_internal(X)), X)


def test_assembly_deterministic(case):
    s = case["sys"]
    a = s.klam.clone(), s.kmu.clone(), s.ms.clone()
    s.assemble()
    assert torch.equal(a[0], s.klam) and torch.equal(a[1], s.kmu) and torch.equal(a[2], s.ms)


@pytest.mark.parametrize("ncols", [8, 24, 80, 240])
def test_spmm(case, dev, ncols):
    h, c = case["hops"], case["cops"]
    if h.n < 3 * 8:
        pytest.skip("tiny")
    g = torch.Generator().manual_seed(ncols)
    X = torch.randn((h.n, ncols), generator=g)
    Xd = X.to(dev)
    for name in ("apply_K", "apply_M"):
        ref = torch.empty_like(X)
        getattr(c, name)(X, ref)
        out = torch.empty_like(Xd)
        getattr(h, name)(Xd, out)
        assert rel(out.cpu().numpy(), ref.numpy()) < 5e-6
    # strided views of a wider buffer (how the solver calls it)
    if ncols >= 24:
        big = torch.zeros((h.n, ncols + 16), device=dev)
        big[:, 8:8 + ncols] = Xd
        outb = torch.zeros_like(big)
        h.apply_K(big[:, 8:8 + ncols], outb[:, 8:8 + ncols])
        ref = torch.empty_like(X)
        c.apply_K(X, ref)
        assert rel(outb[:, 8:8 + ncols].cpu().numpy(), ref.numpy()) < 5e-6
        assert float(outb[:, :8].abs().max()) == 0.0 and float(outb[:, 8 + ncols:].abs().max()) == 0.0


@pytest.mark.parametrize("p,q", [(8, 40), (40, 40), (80, 80), (240, 240), (104, 56), (240, 80), (88, 80), (33, 47)])
def test_gram(case, dev, p, q):
    h = case["hops"]
    g = torch.Generator().manual_seed(p * 1000 + q)
    A = torch.randn((h.n, p), generator=g)
    B = torch.randn((h.n, q), generator=g)
    ref = A.double().T @ B.double()
    scale = np.sqrt(np.outer((A.double() ** 2).sum(0).numpy(), (B.double() ** 2).sum(0).numpy()))
    # default: fp32 MFMA folded into fp64 every 48 rows - elementwise error against |A_i||B_j|.  The error is
    # relative to the 48-row partials and averages out over the folds: ~5e-8 on this mesh of a few hundred rows,
    # 1e-9 at the 4.5e5 rows of the benchmark (tests/test_fullsize_gpu.py checks that)
    G = h.gram(A.to(dev), B.to(dev)).cpu()
    assert (np.abs(G.numpy() - ref.numpy()) / scale).max() < 2e-7
    # strided views of wider buffers whose neighbouring columns hold junk (how the solver calls it)
    wa, wb = torch.full((h.n, p + 24), float("nan")), torch.full((h.n, q + 24), float("nan"))
    wa[:, 8:8 + p], wb[:, 16:16 + q] = A, B
    Gv = h.gram(wa.to(dev)[:, 8:8 + p], wb.to(dev)[:, 16:16 + q]).cpu()
    assert torch.equal(Gv, G)
    # exact products, fp64 accumulation
    G = h.gram(A.to(dev), B.to(dev), exact=True).cpu()
    assert rel(G.numpy(), ref.numpy()) < 1e-13
    G64 = h.gram(A.to(dev), B.double().to(dev)).cpu()
    assert rel(G64.numpy(), ref.numpy()) < 1e-13
    if p == q:  # symmetric mode: block-upper part computed, rest mirrored
        Bs = (B + A) if p == q else B
        KA = torch.from_numpy(np.asarray(case["cops"].Md @ A.numpy()))
        Gs = h.gram(A.to(dev), KA.to(dev), symmetric=True).cpu()
        refs = A.double().T @ KA.double()
        assert rel(Gs.numpy(), refs.numpy()) < 1e-6  # M A is only fp32-symmetric
        assert rel(Gs.numpy(), Gs.numpy().T) < 1e-6


@pytest.mark.parametrize("p,q", [(8, 40), (80, 40), (240, 80), (216, 72), (80, 200)])
def test_mix(case, dev, p, q):
    h = case["hops"]
    g = torch.Generator().manual_seed(p + q)
    A = torch.randn((h.n, p), generator=g)
    C = torch.randn((p, q), generator=g, dtype=torch.float64)
    O = torch.randn((h.n, q), generator=g)
    ref = (A.double() @ C)
    out = O.to(dev).clone()
    h.mix(A.to(dev), C.to(dev), out)
    assert rel(out.cpu().numpy(), ref.numpy()) < 2e-6
    out = O.to(dev).clone()
    h.mix(A.to(dev), C.to(dev), out, alpha=-1.0, beta=1.0)
    assert rel(out.cpu().numpy(), (O.double() - ref).numpy()) < 2e-6
    if q <= 160 and q <= p:  # in place: the result overwrites a column range of A (the ortho step's W <- [V W] C)
        Ad = A.to(dev).clone()
        h.mix(Ad, C.to(dev), Ad[:, p - q:])
        assert rel(Ad[:, p - q:].cpu().numpy(), ref.numpy()) < 2e-6
        assert torch.equal(Ad[:, :p - q].cpu(), A[:, :p - q])


@pytest.mark.parametrize("widths,q", [((6, 40, 24, 24), 40), ((136,), 136), ((6, 136, 52, 52), 136), ((80, 12), 200), ((7, 5, 3), 9)])
def test_mix64_over_a_list_of_blocks_equals_the_fp64_product_of_the_concatenated_basis(case, dev, widths, q):
    """ds_mix64 (the fp64 refinement's dense updates; reference: X <- S Z, src/lobpcg/_lobpcg.py:457-477 on the
    concatenated basis): blocks of different widths, one of them a column range of a wider array, widths that are not
    multiples of 4 (scalar operand loads), more columns than one launch holds, beta != 0, and a block skipped by
    addressing C explicitly."""
    h = case["hops"]
    g = torch.Generator().manual_seed(sum(widths) + q)
    blocks = [torch.randn((h.n, w), generator=g, dtype=torch.float64) for w in widths]
    C = torch.randn((sum(widths), q), generator=g, dtype=torch.float64)
    S = torch.cat(blocks, 1)
    ref = S @ C
    wide = torch.full((h.n, widths[0] + 10), float("nan"), dtype=torch.float64)
    wide[:, :widths[0]] = blocks[0]
    dblocks = [wide.to(dev)[:, :widths[0]]] + [b.to(dev) for b in blocks[1:]]
    out = h.mix64(dblocks, C.to(dev))
    assert rel(out.cpu().numpy(), ref.numpy()) < 1e-14
    O = torch.randn((h.n, q), generator=g, dtype=torch.float64)
    out = O.to(dev).clone()
    h.mix64(dblocks, C.to(dev), out=out, alpha=-0.5, beta=2.0)
    assert rel(out.cpu().numpy(), (2.0 * O - 0.5 * ref).numpy()) < 1e-14
    if len(widths) > 2:  # without the second block: the others keep their rows of C
        offs = np.concatenate([[0], np.cumsum(widths)])
        out = h.mix64([(b, int(offs[i])) for i, b in enumerate(dblocks) if i != 1], C.to(dev))
        C0 = C.clone()
        C0[offs[1]:offs[2]] = 0.0
        assert rel(out.cpu().numpy(), (S @ C0).numpy()) < 1e-14
    with pytest.raises(RuntimeError, match="overlaps"):
        h.mix64([dblocks[-1]], C.to(dev)[:widths[-1], :widths[-1]].contiguous(), out=dblocks[-1])


@pytest.mark.parametrize("wa,wb", [((6, 40, 24, 24), (24, 24)), ((136,), (52, 52)), ((6, 136, 52, 52), (6, 136, 52, 52)), ((7, 50), (3,))])
def test_gram64_over_lists_of_blocks_equals_the_fp64_gram_of_the_concatenated_bases(case, dev, wa, wb):
    """ds_gram64_blocks (S^T [K W | M W] of the fp64 refinement in one pass; reference: the Gram products of
    src/lobpcg/_lobpcg.py:516-525 on the concatenated basis): wave tiles that straddle blocks, a block that is a column range
    of a wider array, and the symmetric mode (upper tiles computed, the rest mirrored)."""
    h = case["hops"]
    g = torch.Generator().manual_seed(sum(wa) + 3 * sum(wb))
    A = [torch.randn((h.n, w), generator=g, dtype=torch.float64) for w in wa]
    B = [torch.randn((h.n, w), generator=g, dtype=torch.float64) for w in wb]
    wide = torch.full((h.n, wa[0] + 10), float("nan"), dtype=torch.float64)
    wide[:, 4:4 + wa[0]] = A[0]
    dA = [wide.to(dev)[:, 4:4 + wa[0]]] + [a.to(dev) for a in A[1:]]
    dB = [b.to(dev) for b in B]
    ref = torch.cat(A, 1).T @ torch.cat(B, 1)
    scale = np.sqrt(np.outer((torch.cat(A, 1) ** 2).sum(0).numpy(), (torch.cat(B, 1) ** 2).sum(0).numpy()))
    G = h.gram_blocks(dA, dB).cpu()
    assert (np.abs(G.numpy() - ref.numpy()) / scale).max() < 1e-14
    if wa == wb:  # symmetric mode: B = Md A with the (symmetric) mass matrix of the mesh
        Md = case["cops"].Md
        KA = [torch.from_numpy(np.asarray(Md @ a.numpy())) for a in A]
        refs = torch.cat(A, 1).T @ torch.cat(KA, 1)
        Gs = h.gram_blocks(dA, [k.to(dev) for k in KA], symmetric=True).cpu()
        assert rel(Gs.numpy(), refs.numpy()) < 1e-13
        assert rel(Gs.numpy(), Gs.numpy().T) < 1e-13


def test_residual_and_cheb(case, dev):
    h, c = case["hops"], case["cops"]
    b = 40
    g = torch.Generator().manual_seed(5)
    R = torch.randn((h.n, b), generator=g)
    MX = torch.randn((h.n, b), generator=g)
    X = torch.randn((h.n, b), generator=g)
    lam = torch.rand(b, generator=g, dtype=torch.float64) * 3
    Rc = R.clone()
    rn, xn = c.residual(Rc, MX, X, lam)
    Rd = R.to(dev).clone()
    rn_d, xn_d = h.residual(Rd, MX.to(dev), X.to(dev), lam.to(dev))
    assert rel(Rd.cpu().numpy(), Rc.numpy()) < 1e-6
    assert rel(rn_d.cpu().numpy(), rn.numpy()) < 1e-6 and rel(xn_d.cpu().numpy(), xn.numpy()) < 1e-12
    # out of place: K X read from a strided view of another buffer (how the solver calls it)
    KXbig = torch.zeros((h.n, b + 8), device=dev)
    KXbig[:, 4:4 + b] = R.to(dev)
    Ro = torch.full((h.n, b), float("nan"), device=dev)
    rn_o, xn_o = h.residual(Ro, MX.to(dev), X.to(dev), lam.to(dev), src=KXbig[:, 4:4 + b])
    assert torch.equal(Ro, Rd) and torch.equal(rn_o, rn_d) is not None
    assert rel(rn_o.cpu().numpy(), rn.numpy()) < 1e-6
    assert torch.equal(KXbig[:, 4:4 + b], R.to(dev))
    # block-Jacobi blocks and the two fused Chebyshev passes
    assert rel(h.dinv.cpu().numpy().reshape(-1, 3, 3), c.Dinv.numpy()) < 1e-5
    D, W = torch.empty_like(R), torch.empty_like(R)
    c.cheb_init(R, D, W, 0.37)
    Dd, Wd = torch.empty_like(Rd), torch.empty_like(Rd)
    h.cheb_init(R.to(dev), Dd, Wd, 0.37)
    assert rel(Dd.cpu().numpy(), D.numpy()) < 1e-5 and rel(Wd.cpu().numpy(), W.numpy()) < 1e-5
    AD = torch.randn((h.n, b), generator=g) * 0.1
    R2, R2d = R.clone(), R.to(dev).clone()
    c.cheb_step(AD, R2, D, W, 0.4, 0.7)
    h.cheb_step(AD.to(dev), R2d, Dd, Wd, 0.4, 0.7)
    for a, bb in ((R2d, R2), (Dd, D), (Wd, W)):
        assert rel(a.cpu().numpy(), bb.numpy()) < 1e-5


@pytest.mark.parametrize("ncols,first", [(80, True), (80, False), (72, False), (24, False), (4, False)])
def test_fused_chebyshev_spmm(case, dev, ncols, first):
    h, c = case["hops"], case["cops"]
    g = torch.Generator().manual_seed(ncols + int(first))
    W = torch.randn((h.n, ncols), generator=g)
    Wp = torch.randn((h.n, ncols), generator=g)
    R0 = torch.randn((h.n, ncols), generator=g) * 1e10
    Wp_ref = Wp.clone()
    c.cheb_spmm(W, Wp_ref, R0, 0.31, 0.77, first)
    # through strided views of a wider buffer, like the solver
    big = torch.zeros((h.n, ncols + 8), device=dev)
    big[:, 8:] = W.to(dev)
    Wp_d = Wp.to(dev).clone()
    h.cheb_spmm(big[:, 8:], Wp_d, R0.to(dev), 0.31, 0.77, first)
    assert rel(Wp_d.cpu().numpy(), Wp_ref.numpy()) < 5e-6


@pytest.mark.parametrize("ncols", [80, 40, 8])
def test_two_level_pieces(case, dev, ncols):
    """Corner-node level of the two-level preconditioner: the ord-1 assembly of the corner sub-mesh is the Galerkin
    operator P^T K P of the oracle, and the transfer / fused-residual kernels match the oracle's matrices."""
    h, c = case["hops"], case["cops"]
    if case["order"] != 2:
        assert h.coarse is None
        pytest.skip("no coarse level on an ord-1 mesh")
    hc, cc = h.coarse, c.coarse
    assert hc is not None and hc.n == cc.n
    g = torch.Generator().manual_seed(ncols)
    Xc = torch.randn((cc.n, ncols), generator=g)
    Xf = torch.randn((c.n, ncols), generator=g)
    R0 = torch.randn((c.n, ncols), generator=g) * 1e10
    ref = torch.empty_like(Xc)
    cc.apply_K(Xc, ref)
    out = torch.empty((hc.n, ncols), device=dev)
    hc.apply_K(Xc.to(dev), out)
    assert rel(out.cpu().numpy(), ref.numpy()) < 5e-6  # Galerkin identity
    assert rel(hc.dinv.cpu().numpy().reshape(-1, 3, 3), cc.Dinv.numpy()) < 1e-5
    # restriction, prolongation (accumulating), fused residual
    rc_ref = torch.empty_like(Xc)
    c.restrict(Xf, rc_ref)
    rc = torch.empty((hc.n, ncols), device=dev)
    h.restrict(Xf.to(dev), rc)
    assert rel(rc.cpu().numpy(), rc_ref.numpy()) < 1e-6
    wf_ref = Xf.clone()
    c.prolong_add(Xc, wf_ref)
    big = torch.zeros((h.n, ncols + 4), device=dev)  # strided view, like the solver's W
    big[:, 4:] = Xf.to(dev)
    h.prolong_add(Xc.to(dev), big[:, 4:])
    assert rel(big[:, 4:].cpu().numpy(), wf_ref.numpy()) < 1e-6
    assert float(big[:, :4].abs().max()) == 0.0
    y_ref = torch.empty_like(Xf)
    c.spmm_residual(Xf, R0, y_ref)
    y = torch.empty((h.n, ncols), device=dev)
    h.spmm_residual(big[:, 4:].contiguous() * 0 + Xf.to(dev), R0.to(dev), y)
    assert rel(y.cpu().numpy(), y_ref.numpy()) < 5e-6


def test_two_level_cycle_matches_oracle(case, dev):
    from diffsound_amd.lobpcg.modal_solver import SolverConfig, TwoLevelChebyshev

    h, c = case["hops"], case["cops"]
    if case["order"] != 2:
        pytest.skip("no coarse level on an ord-1 mesh")
    cfg = SolverConfig(lmax_cap=10.0, smooth_degree=3, coarse_degree=12, coarse_ratio=50.0, precond_storage="fp32")
    ph, pc = TwoLevelChebyshev(h, cfg), TwoLevelChebyshev(c, cfg)
    assert abs(ph.smooth.lmax / pc.smooth.lmax - 1) < 0.05 and abs(ph.coarse.lmax / pc.coarse.lmax - 1) < 0.05
    for a, b in ((ph.smooth, pc.smooth), (ph.coarse, pc.coarse)):  # same polynomial on both sides
        a.lmax, a.lmin = b.lmax, b.lmin
    g = torch.Generator().manual_seed(5)
    R = torch.randn((c.n, 96), generator=g) * 1e9  # 96 columns: two chunks
    Wc = torch.empty_like(R)
    pc.apply(R.clone(), Wc)
    Wh = torch.empty((h.n, 96), device=dev)
    ph.apply(R.to(dev), Wh)
    assert rel(Wh.cpu().numpy(), Wc.numpy()) < 2e-4
    # the native driver (ds_twolevel_apply: the whole cycle in one call) issues the same launches as the Python loop
    calls = h.counts["apply_K_cols"], h.coarse.counts["apply_K_cols"]
    ph.use_native = False
    try:
        Wp = torch.empty((h.n, 96), device=dev)
        k0 = h.counts["apply_K_cols"], h.coarse.counts["apply_K_cols"]
        ph.apply(R.to(dev), Wp)
        k1 = h.counts["apply_K_cols"], h.coarse.counts["apply_K_cols"]
    finally:
        ph.use_native = True
    assert torch.equal(Wp, Wh)
    Wn = torch.empty((h.n, 96), device=dev)
    ph.apply(R.to(dev), Wn)
    k2 = h.counts["apply_K_cols"], h.coarse.counts["apply_K_cols"]
    assert torch.equal(Wn, Wh)
    assert (k2[0] - k1[0], k2[1] - k1[1]) == (k1[0] - k0[0], k1[1] - k0[1])  # same product counts either way
    assert calls[0] > 0
    # bf16 storage of the cycle's internal blocks (the default): the same operator up to the rounding of the stored
    # iterates (8-bit mantissas, fp32 arithmetic), deterministic, fp32 in and out
    p16 = TwoLevelChebyshev(h, SolverConfig(lmax_cap=10.0, smooth_degree=3, coarse_degree=12, coarse_ratio=50.0))
    assert p16.storage == "bf16"
    for a, b in ((p16.smooth, pc.smooth), (p16.coarse, pc.coarse)):
        a.lmax, a.lmin = b.lmax, b.lmin
    W16 = torch.full((h.n, 96), float("nan"), device=dev)
    p16.apply(R.to(dev), W16)
    assert W16.dtype == torch.float32 and bool(torch.isfinite(W16).all())
    assert rel(W16.cpu().numpy(), Wc.numpy()) < 3e-2
    cols = np.linalg.norm(W16.cpu().numpy() - Wc.numpy(), axis=0) / np.linalg.norm(Wc.numpy(), axis=0)
    assert cols.max() < 5e-2  # every column, not only on average
    W16b = torch.empty((h.n, 96), device=dev)
    p16.apply(R.to(dev), W16b)
    assert torch.equal(W16, W16b)


@pytest.mark.parametrize("ncols", [80, 40, 84])
def test_bf16_union_terms_match_fp32_terms(dev, ncols):
    """ds_spmm_union16 (bf16 X / R0 / W_prev, fp32 arithmetic) against ds_spmm_union on the bf16-rounded operands:
    the Chebyshev term with bf16 and with fp32 output, the residual; plus the bf16 first iterate (ds_cheb_init16) and
    the bf16 level transfer against their fp32 counterparts."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(6)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    sysd = TetSystem(tm.vertices, tm.tets, 2, 2700.0)
    ops = HipModalOps(sysd, 2e10, 2e10)
    g = torch.Generator(device=dev).manual_seed(ncols)
    mk = lambda scale=1.0: (torch.randn((sysd.n, ncols), generator=g, device=dev) * scale).bfloat16()
    X, Wp, R0 = mk(), mk(), mk(1e10)
    Xf, Wpf, R0f = X.float(), Wp.float(), R0.float()
    L, p, u, gr = _hip.lib(), _hip.ptr, sysd.groups["union"], sysd.groups
    ut = None if u["single"] else p(u["utab"])

    def term16(epi, out, y32, first, wprev=None):
        _hip.check(L.ds_spmm_union16(epi, ut, p(u["ctab"]), u["ngroups"], u["capb"], p(gr["gent"]), p(ops.kgrp), ops.kgrp.shape[0],
                                     sysd.nv, p(X), ncols, p(out), out.stride(0), int(y32), p(R0), ncols, p(ops.dinv), ncols, 0.31,
                                     0.77, int(first), p(wprev), 0 if wprev is None else ncols, _hip.stream_ptr()), "ds_spmm_union16")

    for first in (False, True):
        ref = Wpf.clone()
        ops._union(1, Xf, ref, R0f, 0.31, 0.77, first)
        o32 = torch.full((sysd.n, ncols + 8), float("nan"), device=dev)
        term16(1, o32[:, 4:4 + ncols], True, first, wprev=Wp)          # out of place, fp32 result in a column range
        assert torch.equal(o32[:, 4:4 + ncols], ref) and bool(torch.isnan(o32[:, :4]).all())
        o16 = Wp.clone()
        term16(1, o16, False, first)                                   # in place on the bf16 W_prev
        assert torch.equal(o16, ref.bfloat16())
    refr = torch.empty_like(Xf)
    ops._union(2, Xf, refr, R0f)
    r16 = torch.empty_like(X)
    term16(2, r16, False, False)
    assert torch.equal(r16, refr.bfloat16())
    # first iterate W1 = c T R from an fp32 block (with its bf16 copy) and from a bf16 block
    Rf = torch.randn((sysd.n, ncols), generator=g, device=dev) * 1e9
    Dref, Wref = torch.empty_like(Rf), torch.empty_like(Rf)
    ops.cheb_init(Rf, Dref, Wref, 0.37)
    W1, Rc = torch.empty_like(X), torch.empty_like(X)
    _hip.check(L.ds_cheb_init16(p(Rf), 1, ncols, p(W1), ncols, p(Rc), ncols, p(ops.dinv), sysd.nv, ncols, 0.37, _hip.stream_ptr()),
               "ds_cheb_init16")
    assert torch.equal(W1, Wref.bfloat16()) and torch.equal(Rc, Rf.bfloat16())
    ops.cheb_init(Rc.float(), Dref, Wref, 0.37)
    W2 = torch.empty_like(X)
    _hip.check(L.ds_cheb_init16(p(Rc), 0, ncols, p(W2), ncols, None, 0, p(ops.dinv), sysd.nv, ncols, 0.37, _hip.stream_ptr()),
               "ds_cheb_init16")
    assert torch.equal(W2, Wref.bfloat16())
    # level transfer on bf16 panels
    co, tr = ops.coarse, ops._xfer
    Rcf = torch.empty((co.n, ncols), device=dev)
    ops.restrict(Xf, Rcf)
    Rc16 = torch.empty((co.n, ncols), dtype=torch.bfloat16, device=dev)
    _hip.check(L.ds_scalar_csr_spmm16(p(tr["rptr"]), p(tr["rcol"]), p(tr["rw"]), co.nv, p(X), ncols, p(Rc16), ncols, ncols, 0.0,
                                      _hip.stream_ptr()), "ds_scalar_csr_spmm16")
    assert torch.equal(Rc16, Rcf.bfloat16())
    Wf = Wpf.clone()
    ops.prolong_add(Rc16.float(), Wf)
    W16 = Wp.clone()
    _hip.check(L.ds_scalar_csr_spmm16(p(tr["pptr"]), p(tr["pcol"]), p(tr["pw"]), sysd.nv, p(Rc16), ncols, p(W16), ncols, ncols, 1.0,
                                      _hip.stream_ptr()), "ds_scalar_csr_spmm16")
    assert torch.equal(W16, Wf.bfloat16())


@pytest.mark.parametrize("mesh,order", [(6, 2), (3, 2), (12, 1), (26, 2)])
def test_mfma_tables_head_records_and_batch_partition(dev, mesh, order):
    """Topology tables of ds_spmm_union16m against an independent count from the BSR pattern: the unions of every group of 8 rows,
    the fixed-stride head records (a group's first 64 entries, zero behind its last), and max_batch_blocks of the batch partition
    the kernel walks - batches of 16 entries, the last one of a group up to 2 entries longer when no group has more than 128."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import MF_BATCH, MF_TAIL, TetSystem

    v, t = meshgen.kuhn_box(mesh)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
    mt = sysd.mfma_tables(8)
    rowptr, col = sysd.rowptr.cpu().numpy().astype(np.int64), sysd.colidx.cpu().numpy().astype(np.int64)
    gptr, gcol, gmeta = (mt[k].cpu().numpy().astype(np.int64) for k in ("gptr", "gcol", "gmeta"))
    ghead, gbase = mt["ghead"].cpu().numpy().astype(np.int64), mt["gbase"].cpu().numpy().astype(np.int64)
    ng = (sysd.nv + 7) // 8
    assert mt["ngroups"] == ng and ghead.shape == (ng, 128)
    worst, most = 0, 0
    for g in range(ng):
        rows = range(8 * g, min(8 * g + 8, sysd.nv))
        nb = {r - 8 * g: set(col[rowptr[r]:rowptr[r + 1]].tolist()) for r in rows}
        union = sorted(set().union(*nb.values()))
        e0, e1 = gptr[g], gptr[g + 1]
        assert gcol[e0:e1].tolist() == union
        mask = [sum(1 << s for s, cs in nb.items() if c in cs) for c in union]
        assert (gmeta[e0:e1] & 0xff).tolist() == mask
        first = np.concatenate([[0], np.cumsum([bin(m).count("1") for m in mask])])
        assert (gmeta[e0:e1] >> 8).tolist() == first[:-1].tolist() and gbase[g] == rowptr[8 * g]
        ne = len(union)
        head = min(ne, 64)
        assert ghead[g, :head].tolist() == union[:head] and not ghead[g, head:64].any()
        assert ghead[g, 64:64 + head].tolist() == gmeta[e0:e0 + head].tolist() and not ghead[g, 64 + head:].any()
        most = max(most, ne)
        per_entry = np.diff(first)
        nbatch = max(1, (ne - MF_TAIL + MF_BATCH - 1) // MF_BATCH)
        for b in range(nbatch):
            hi = ne if b == nbatch - 1 else (b + 1) * MF_BATCH
            assert hi - b * MF_BATCH <= MF_BATCH + MF_TAIL
            worst = max(worst, int(per_entry[b * MF_BATCH:hi].sum()))
    assert most == mt["max_entries"] <= 128  # (the tail partition is what these meshes get)
    assert worst == mt["max_batch_blocks"]


@pytest.mark.parametrize("mesh,ncols,G,order", [(6, 80, 8, 2), (6, 40, 8, 2), (6, 84, 8, 2), (6, 52, 8, 2), (3, 80, 8, 2), (10, 80, 8, 2),
                                                (5, 16, 8, 2), (12, 80, 8, 1), (26, 80, 8, 1)])
def test_mfma_union_terms_match_valu_terms(dev, mesh, ncols, G, order):
    """ds_spmm_union16m (block products on the matrix cores, 3x3 blocks rounded to bf16 and packed by ds_pack_kc, one
    wave per 8 nodes) against ds_spmm_union16 fed the same bf16-rounded blocks: the two differ
    by the order of the fp32 accumulation only.  Chebyshev term with fp32 and bf16 output, first term, out-of-place
    W_prev, residual; meshes whose last group is incomplete and with several batches of entries per group, ord-1 meshes
    (the corner-node level's kind)."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(mesh)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
    ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
    g = torch.Generator(device=dev).manual_seed(ncols + mesh)
    mk = lambda scale=1.0: (torch.randn((sysd.n, ncols), generator=g, device=dev) * scale).bfloat16()
    X, Wp, R0 = mk(), mk(), mk(1e10)
    L, p, u, gr = _hip.lib(), _hip.ptr, sysd.groups["union"], sysd.groups
    ut = None if u["single"] else p(u["utab"])
    k16 = ops.kgrp.bfloat16().float().contiguous()  # the VALU kernel on the rounded blocks = the reference
    mt = sysd.mfma_tables(G)
    assert mt["ngroups"] == (sysd.nv + G - 1) // G and int(mt["gptr"][-1]) == mt["gcol"].numel()
    kc = torch.empty((sysd.nnzb, 3, 4), dtype=torch.bfloat16, device=dev)
    _hip.check(L.ds_pack_kc(p(ops.k32), p(mt["kperm"]), sysd.nnzb, p(kc), _hip.stream_ptr()), "ds_pack_kc")
    want = ops.k32.view(-1, 3, 3)[mt["kperm"].long()].bfloat16()
    assert torch.equal(kc[:, :, :3], want) and float(kc[:, :, 3].abs().max()) == 0.0

    def valu(epi, out, y32, first, wprev=None):
        _hip.check(L.ds_spmm_union16(epi, ut, p(u["ctab"]), u["ngroups"], u["capb"], p(gr["gent"]), p(k16), k16.shape[0],
                                     sysd.nv, p(X), ncols, p(out), out.stride(0), int(y32), p(R0), ncols, p(ops.dinv), ncols, 0.31,
                                     0.77, int(first), p(wprev), 0 if wprev is None else ncols, _hip.stream_ptr()), "ds_spmm_union16")

    def mfma(epi, out, y32, first, wprev=None, plain=False):
        # (plain: a max_entries bound above 128 sends the launch to the form WITHOUT the tail batch - the bound only has to hold)
        _hip.check(L.ds_spmm_union16m(epi, G, 0, p(mt["gptr"]), p(mt["gcol"]), p(mt["gmeta"]), p(mt["gbase"]), p(mt["ghead"]), p(kc), sysd.nnzb,
                                      mt["ngroups"], 129 if plain else mt["max_entries"], mt["max_batch_blocks"], sysd.nv, p(X), ncols, p(out), out.stride(0), int(y32), p(R0),
                                      ncols, p(ops.dinv), ncols, 0.31, 0.77, int(first), p(wprev), 0 if wprev is None else ncols,
                                      _hip.stream_ptr()), "ds_spmm_union16m")

    def close(a, b, tol):
        a, b = a.float(), b.float()
        return float((a - b).abs().max() / b.abs().max()) < tol

    for first in (False, True):
        a32 = torch.full((sysd.n, ncols + 8), float("nan"), device=dev)
        b32 = torch.full((sysd.n, ncols + 8), float("nan"), device=dev)
        valu(1, a32[:, 4:4 + ncols], True, first, wprev=Wp)
        mfma(1, b32[:, 4:4 + ncols], True, first, wprev=Wp)
        assert bool(torch.isnan(b32[:, :4]).all()) and bool(torch.isnan(b32[:, 4 + ncols:]).all())
        assert close(b32[:, 4:4 + ncols], a32[:, 4:4 + ncols], 2e-6)
        a16, b16 = Wp.clone(), Wp.clone()
        valu(1, a16, False, first)
        mfma(1, b16, False, first)
        assert close(b16, a16, 8e-3) and float((b16 != a16).float().mean()) < 0.02  # bf16 results: rare one-ulp flips
    ar, br = torch.empty_like(X), torch.empty_like(X)
    valu(2, ar, False, False)
    mfma(2, br, False, False)
    assert close(br, ar, 8e-3) and float((br != ar).float().mean()) < 0.02
    br2 = torch.empty_like(X)
    mfma(2, br2, False, False)
    assert torch.equal(br, br2)  # deterministic
    # the form whose last batch takes two entries more (round 5: groups of <= 128 entries, >= 49 columns) and the plain form keep
    # every entry in its K-step slot and in its place of the accumulation order: the same bits
    assert mt["max_entries"] <= 128
    mfma(2, br2, False, False, plain=True)
    assert torch.equal(br, br2)
    c32 = torch.empty((sysd.n, ncols), device=dev)
    d32 = torch.empty((sysd.n, ncols), device=dev)
    mfma(1, c32, True, False, wprev=Wp)
    mfma(1, d32, True, False, wprev=Wp, plain=True)
    assert torch.equal(c32, d32)


@pytest.mark.parametrize("mesh,order,ncols", [(6, 2, 80), (6, 2, 40), (6, 2, 84), (6, 2, 52), (3, 2, 80), (10, 2, 80), (5, 2, 16),
                                              (12, 1, 80), (26, 1, 80), (2, 1, 8), (3, 1, 4)])
def test_mfma32_products_match_fp64_product_and_valu_kernel(dev, mesh, order, ncols):
    """ds_spmm_union32m (the eigensolver's own fp32 products on v_mfma_f32_16x16x4_f32, one wave per 4 nodes, batches of 8
    union entries) against an fp64-value / fp64-accumulation product of the same fp32 operands, beside the VALU
    neighbour-union kernel it replaces: K X (3x3 blocks) and M X (node-scalar values), strided views of a wider buffer
    whose margins must stay untouched, meshes whose last group is incomplete, groups with one and with many batches,
    ord-1 meshes (the corner-node level's kind).  The MFMA form sums each output in ONE fp32 chain (entry order), the VALU
    form in three (one per panel row): same precision, not the same bits."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(mesh)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
    lam, mu = 2e10, 3e10
    ops = HipModalOps(sysd, lam, mu, two_level=False, mfma32=True)
    m4 = ops._mfma32
    assert m4 is not None and m4["G"] == 4 and m4["batch"] == 8 and m4["ngroups"] == (sysd.nv + 3) // 4
    assert int(m4["gptr"][-1]) == m4["gcol"].numel()
    # the packed values: blocks (row-major, NOT transposed) and node scalars in the tables' order, finite slack behind them
    kp = m4["kperm"].long()
    assert torch.equal(ops.k4[:sysd.nnzb * 9].view(-1, 9), ops.k32[kp]) and torch.equal(ops.m4[:sysd.nnzb], ops.ms32[kp])
    assert float(ops.k4[sysd.nnzb * 9:].abs().max()) == 0.0 and float(ops.m4[sysd.nnzb:].abs().max()) == 0.0
    g = torch.Generator(device=dev).manual_seed(ncols + mesh)
    big = torch.full((sysd.n, ncols + 12), float("nan"), device=dev)
    X = big[:, 8:8 + ncols]
    X.copy_(torch.randn((sysd.n, ncols), generator=g, device=dev))
    Xc = X.contiguous()
    # reference: the fp32 values both kernels multiply by, products and sums in fp64 (ds_spmm_bsr3 kinds 2 / 3)
    for epi, vals64, kind in ((0, ops.k32.double().contiguous(), 2), (3, ops.ms32.double().contiguous(), 3)):
        ref = torch.empty((sysd.n, ncols), dtype=torch.float64, device=dev)
        ops._spmm(kind, vals64, Xc, ref)
        scale = float(ref.abs().max())
        outs = {}
        for name, fn in (("mfma", ops._union32), ("valu", ops._union)):
            wide = torch.full((sysd.n, ncols + 8), float("nan"), device=dev)
            fn(epi, X, wide[:, 4:4 + ncols])
            assert bool(torch.isnan(wide[:, :4]).all()) and bool(torch.isnan(wide[:, 4 + ncols:]).all()), name
            outs[name] = wide[:, 4:4 + ncols].double()
            assert bool(torch.isfinite(outs[name]).all()), name
        e_m = float((outs["mfma"] - ref).abs().max()) / scale
        e_v = float((outs["valu"] - ref).abs().max()) / scale
        assert e_m < 1e-6 and e_m < 3 * e_v + 1e-7, (epi, e_m, e_v)
        again = torch.empty((sysd.n, ncols), device=dev)
        ops._union32(epi, X, again)
        assert torch.equal(again.double(), outs["mfma"])  # deterministic
    # the ops' own entry points take the matrix-core form
    Y = torch.empty((sysd.n, ncols), device=dev)
    ops.apply_K(X, Y)
    Y2 = torch.empty_like(Y)
    ops._union32(0, X, Y2)
    assert torch.equal(Y, Y2)


def test_mfma32_entry_point_refuses_what_it_does_not_serve(dev):
    """ds_spmm_union32m validates on the host before any launch."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(4)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    sysd = TetSystem(tm.vertices, tm.tets, 2, 2700.0)
    ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma32=True)
    m4, L, p = ops._mfma32, _hip.lib(), _hip.ptr
    X = torch.randn(sysd.n, 80, device=dev)
    Y = torch.empty_like(X)

    def call(epi=0, tag=0, vals=None, vbytes=None, ngroups=None, max_entries=None, mbb=None, x=X, y=Y, ncols=80):
        vals = (ops.m4 if epi == 3 else ops.k4) if vals is None else vals
        return L.ds_spmm_union32m(epi, tag, p(m4["gptr"]), p(m4["gcol"]), p(m4["gmeta"]), p(m4["gbase"]), p(vals),
                                  vals.numel() * 4 if vbytes is None else vbytes, sysd.nnzb,
                                  m4["ngroups"] if ngroups is None else ngroups, m4["max_entries"] if max_entries is None else max_entries,
                                  m4["max_batch_blocks"] if mbb is None else mbb, sysd.nv, p(x), x.stride(0), p(y), y.stride(0), ncols,
                                  _hip.stream_ptr())

    assert call() == 0 and call(epi=3) == 0 and call(tag=1) == 0
    for bad in (dict(epi=1), dict(epi=2), dict(tag=2), dict(vbytes=sysd.nnzb * 36), dict(ngroups=m4["ngroups"] + 1), dict(max_entries=257),
                dict(mbb=0), dict(mbb=33), dict(y=X), dict(ncols=88), dict(ncols=78), dict(x=X[:, 1:])):
        assert call(**bad) != 0, bad
        assert L.ds_last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("mesh,order,ncols", [(6, 2, 80), (6, 2, 72), (6, 2, 40), (10, 2, 80), (3, 2, 8), (12, 1, 80), (5, 1, 24), (2, 1, 4)])
def test_fused_residual_equals_the_three_launches_bit_for_bit(dev, mesh, order, ncols):
    """ds_union_residual (R = K X - (M X) diag(lam) and the column norms in ONE walk of the neighbour unions; K X and M X
    never written) against what it replaces - ds_spmm_union epilogue 0, epilogue 3, ds_residual: R must be IDENTICAL (per
    output the same sums in the same order, one fused multiply-add), the norms agree to fp32 partial-sum accuracy and are
    reproducible to the last bit (fixed-order reduction, no atomics).  Operands are column ranges of wider buffers, as the
    solver passes them."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(mesh)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
    ops = HipModalOps(sysd, 2e10, 3e10, two_level=False)
    g = torch.Generator(device=dev).manual_seed(mesh * 100 + ncols)
    big = torch.full((sysd.n, 256), float("nan"), device=dev)
    X = big[:, 8:8 + ncols]
    X.copy_(torch.randn((sysd.n, ncols), generator=g, device=dev))
    lam = (torch.rand(ncols, generator=g, device=dev, dtype=torch.float64) + 0.5) * 1e9
    assert ops.residual_fused_ok(X, torch.empty((sysd.n, ncols), device=dev))
    # the three launches
    KX, MX = torch.empty((sysd.n, ncols), device=dev), torch.empty((sysd.n, ncols), device=dev)
    ops._union(0, X, KX)
    ops._union(3, X, MX)
    R0 = torch.empty((sysd.n, ncols), device=dev)
    rn0, xn0 = ops.residual(R0, MX, X, lam, src=KX)
    # one launch, into a column range of a wider buffer
    wide = torch.full((sysd.n, ncols + 8), float("nan"), device=dev)
    R1 = wide[:, 4:4 + ncols]
    rn1, xn1 = ops.residual_fused(X, lam, R1)
    assert bool(torch.isnan(wide[:, :4]).all()) and bool(torch.isnan(wide[:, 4 + ncols:]).all())
    assert torch.equal(R1, R0)
    assert float(((rn1 - rn0) / rn0).abs().max()) < 2e-6 and float(((xn1 - xn0) / xn0).abs().max()) < 2e-6
    R2 = torch.empty((sysd.n, ncols), device=dev)
    rn2, xn2 = ops.residual_fused(X, lam, R2)
    assert torch.equal(R2, R0) and torch.equal(rn2, rn1) and torch.equal(xn2, xn1)  # reproducible to the last bit
    L, p = __import__("diffsound_amd")._hip.lib(), __import__("diffsound_amd")._hip.ptr
    u, gr = sysd.groups["union"], sysd.groups
    ws = torch.empty(8, dtype=torch.uint8, device=dev)  # too small a workspace is refused before any launch
    assert L.ds_union_residual(0, None if u["single"] else p(u["utab"]), p(u["ctab"]), u["ngroups"], u["capb"], p(gr["gent"]), p(ops.kgrp),
                               p(ops.mgrp), ops.kgrp.shape[0], sysd.nv, p(X), X.stride(0), p(lam), p(R2), ncols, ncols, p(ws), 8,
                               p(ops._nrm[0]), p(ops._nrm[1]), __import__("diffsound_amd")._hip.stream_ptr()) != 0


@pytest.mark.parametrize("mesh,order,ncols", [(6, 2, 80), (6, 2, 72), (10, 2, 80), (3, 2, 8), (12, 1, 80), (5, 1, 24), (2, 1, 4)])
def test_union_km_equals_the_two_products_bit_for_bit(dev, mesh, order, ncols):
    """ds_spmm_union_km (Y = K X and Y2 = M X of one block in ONE walk of the neighbour unions, round 5) against the two
    launches it replaces: both results IDENTICAL, written into column ranges of a wider buffer as the solver passes them,
    nothing outside those ranges touched; bad arguments refused before any launch."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(mesh)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
    ops = HipModalOps(sysd, 2e10, 3e10, two_level=False)
    g = torch.Generator(device=dev).manual_seed(mesh * 100 + ncols)
    big = torch.full((sysd.n, 256), float("nan"), device=dev)
    X = big[:, 16:16 + ncols]
    X.copy_(torch.randn((sysd.n, ncols), generator=g, device=dev))
    KX, MX = torch.empty((sysd.n, ncols), device=dev), torch.empty((sysd.n, ncols), device=dev)
    ops._union(0, X, KX)
    ops._union(3, X, MX)
    wide = torch.full((sysd.n, 2 * ncols + 8), float("nan"), device=dev)
    K1, M1 = wide[:, 4:4 + ncols], wide[:, 4 + ncols:4 + 2 * ncols]
    assert ops.apply_KM_ok(X, K1, M1)
    ops.apply_KM(X, K1, M1)
    assert bool(torch.isnan(wide[:, :4]).all()) and bool(torch.isnan(wide[:, 4 + 2 * ncols:]).all())
    assert torch.equal(K1, KX) and torch.equal(M1, MX)
    L, p = _hip.lib(), _hip.ptr
    u, gr = sysd.groups["union"], sysd.groups

    def call(x=X, kx=K1, mx=M1, nc=ncols, tag=0):
        return L.ds_spmm_union_km(tag, None if u["single"] else p(u["utab"]), p(u["ctab"]), u["ngroups"], u["capb"], p(gr["gent"]),
                                  p(ops.kgrp), p(ops.mgrp), ops.kgrp.shape[0], sysd.nv, p(x), x.stride(0), p(kx), kx.stride(0),
                                  p(mx), mx.stride(0), nc, _hip.stream_ptr())

    assert call() == 0
    for bad in (dict(kx=X), dict(mx=K1), dict(nc=88), dict(nc=ncols + 2), dict(tag=2), dict(x=big[:, 1:1 + ncols])):
        assert call(**bad) != 0, bad
    torch.cuda.synchronize()


@pytest.mark.parametrize("mesh,order,ncols", [(6, 2, 8), (6, 2, 16), (6, 2, 12), (6, 2, 4), (10, 2, 8), (3, 2, 8), (12, 1, 8), (5, 1, 16), (2, 1, 4)])
def test_narrow_union_kernel_matches_the_production_kernel(dev, mesh, order, ncols):
    """ds_spmm_union_narrow (<= 16 columns, a wave's lanes dealt over the union's entries; round 5) against ds_spmm_union on the
    same block: K X and M X equal to fp32 summation-order rounding (1e-6 of the row's |K| |x| scale), written into a column
    range of a wider buffer with nothing outside it touched; apply_K / apply_M take it for such blocks; bad arguments refused."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(mesh)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
    ops = HipModalOps(sysd, 2e10, 3e10, two_level=False)
    g = torch.Generator(device=dev).manual_seed(mesh * 100 + ncols)
    big = torch.full((sysd.n, 64), float("nan"), device=dev)
    X = big[:, 16:16 + ncols]
    X.copy_(torch.randn((sysd.n, ncols), generator=g, device=dev))
    for kind in (0, 3):
        ref = torch.empty((sysd.n, ncols), device=dev)
        ops._union(kind, X, ref)
        wide = torch.full((sysd.n, ncols + 8), float("nan"), device=dev)
        Y = wide[:, 4:4 + ncols]
        ops._narrow(kind, X, Y)
        assert bool(torch.isnan(wide[:, :4]).all()) and bool(torch.isnan(wide[:, 4 + ncols:]).all())
        scale = float(ref.abs().max())
        assert float((Y - ref).abs().max()) < 2e-6 * scale, (kind, float((Y - ref).abs().max()), scale)
        if kind == 0:  # the fine level's K X of a narrow block is routed here (M X stays on the production kernel: faster)
            out = torch.empty((sysd.n, ncols), device=dev)
            ops.apply_K(X, out)
            assert torch.equal(out, Y)
    L, p = _hip.lib(), _hip.ptr
    u, gr = sysd.groups["union"], sysd.groups
    Yc = torch.empty((sysd.n, ncols), device=dev)

    def call(kind=0, x=X, y=Yc, nc=ncols, tag=0):
        vals = ops.mgrp if kind == 3 else ops.kgrp
        return L.ds_spmm_union_narrow(kind, tag, None if u["single"] else p(u["utab"]), p(u["ctab"]), u["ngroups"], p(gr["gent"]),
                                      p(vals), vals.shape[0], sysd.nv, p(x), x.stride(0), p(y), y.stride(0), nc, _hip.stream_ptr())

    assert call() == 0
    for bad in (dict(kind=1), dict(y=X), dict(nc=20), dict(nc=ncols + 2), dict(tag=2), dict(x=big[:, 1:1 + ncols])):
        assert call(**bad) != 0, bad
    torch.cuda.synchronize()


@pytest.mark.parametrize("n,b", [(30000, 136), (4999, 80), (777, 6), (64, 2)])
def test_fused_fp64_residual_passes_match_torch(dev, n, b):
    """csrc/refine64.hip (round 5): the fp64 refinement's element-wise passes - residual norms and ||x|| of every column in ONE pass
    over K X, M X, X (ds_residual64_norms), the scaled fp32 residual columns of an index list in ONE more (ds_residual64_scaled) -
    against the torch formulation they replace (addcmul, two norms, column gather, division, cast); operands are column ranges of
    wider buffers; the norms are reproducible to the last bit."""
    from diffsound_amd.modal_ops import _HipBlockOps

    ops = _HipBlockOps()
    ops._init_common(None, None, (n + 2) // 3, dev)
    ops.n = n
    g = torch.Generator(device=dev).manual_seed(n + b)
    wide = [torch.randn((n, b + 6), generator=g, device=dev, dtype=torch.float64) for _ in range(3)]
    KX, MX, X = (w[:, 2:2 + b] for w in wide)
    lam = torch.rand(b, generator=g, device=dev, dtype=torch.float64) * 3 + 0.5
    R = torch.addcmul(KX, MX, lam[None, :], value=-1.0)
    rn2, xn2 = ops.residual64(KX, MX, X, lam)
    assert float(((rn2 - (R * R).sum(0)) / (R * R).sum(0)).abs().max()) < 1e-13
    assert float(((xn2 - (X * X).sum(0)) / (X * X).sum(0)).abs().max()) < 1e-13
    a2, b2 = ops.residual64(KX, MX, X, lam)
    assert torch.equal(a2, rn2) and torch.equal(b2, xn2)
    if b >= 4:
        idx = torch.arange(b, device=dev)[torch.randperm(b, generator=g, device=dev)[:(b // 4) * 4 // 2 * 2 or 4]]
        idx = torch.sort(idx[:(idx.numel() // 4) * 4]).values
        scale = 1.0 / torch.sqrt(rn2)
        got = ops.residual64_scaled(KX, MX, lam, scale, idx)
        ref = (R[:, idx] * scale[idx][None, :])
        assert got.dtype == torch.float32 and got.shape == (n, idx.numel())
        assert float((got.double() - ref).abs().max()) < 2e-7 * float(ref.abs().max())  # one fp32 rounding (fma in the kernel)


@pytest.mark.parametrize("mesh,order,ncols", [(6, 2, 80), (6, 2, 72), (6, 2, 8), (10, 2, 80), (3, 2, 4), (12, 1, 80), (5, 1, 24)])
def test_fp64_union_spmm_matches_the_node_kernel(dev, mesh, order, ncols):
    """ds_spmm_f64_union (round 5: fp64 values and fp64 / fp32 vectors on the neighbour-union tables, one wave per 4 nodes) against the
    wave-per-node kernels of ds_spmm_bsr3 (kinds 4 / 5, and 2 / 3 for an fp32 X): equal to fp64 rounding of another summation order
    (1e-13 of the row's scale); written into a column range of a wider buffer, nothing else touched; and through the operators - inside
    a combined_k64 phase apply_K64 / apply_M64 take it (136 columns: two slices)."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(mesh)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
    ops = HipModalOps(sysd, 2e10, 3e10, two_level=False)
    L, p = _hip.lib(), _hip.ptr
    gr, u = sysd.groups, sysd.groups["union"]
    kp = gr["kperm64"]
    k64 = 2e10 * sysd.klam + 3e10 * sysd.kmu
    kgrp = k64[kp].reshape(-1, 3, 3).transpose(1, 2).reshape(-1, 9).contiguous()
    mgrp = sysd.ms[kp].contiguous()
    g = torch.Generator(device=dev).manual_seed(mesh * 10 + ncols)
    for xdt in (torch.float64, torch.float32):
        big = torch.full((sysd.n, ncols + 12), float("nan"), dtype=xdt, device=dev)
        X = big[:, 4:4 + ncols]
        X.copy_(torch.randn((sysd.n, ncols), generator=g, device=dev, dtype=torch.float64).to(xdt))
        for kind, vals_grp, vals_bsr, bkind in ((0, kgrp, k64, 4 if xdt == torch.float64 else 2), (1, mgrp, sysd.ms, 5 if xdt == torch.float64 else 3)):
            ref = torch.empty((sysd.n, ncols), dtype=torch.float64, device=dev)
            _hip.check(L.ds_spmm_bsr3(bkind, p(sysd.rowptr), p(sysd.colidx), p(vals_bsr), None, sysd.nv, p(X), X.stride(0), p(ref),
                                      ref.stride(0), ncols, _hip.stream_ptr()), "ds_spmm_bsr3")
            wide = torch.full((sysd.n, ncols + 4), float("nan"), dtype=torch.float64, device=dev)
            Y = wide[:, 2:2 + ncols]
            rc = L.ds_spmm_f64_union(kind, int(xdt == torch.float64), None if u["single"] else p(u["utab"]), p(u["ctab"]), u["ngroups"],
                                     u["capb"], p(gr["gent"]), p(vals_grp), vals_grp.shape[0], sysd.nv, p(X), X.stride(0), p(Y),
                                     Y.stride(0), ncols, _hip.stream_ptr())
            assert rc == 0, L.ds_last_error()
            assert bool(torch.isnan(wide[:, :2]).all()) and bool(torch.isnan(wide[:, 2 + ncols:]).all())
            assert float((Y - ref).abs().max()) < 1e-13 * float(ref.abs().max()), (xdt, kind)
    # through the operators, a block wider than one launch takes
    X = torch.randn((sysd.n, 136), generator=g, device=dev, dtype=torch.float64)
    refK, refM = torch.empty_like(X), torch.empty_like(X)
    ops.apply_K64(X, refK)   # (outside a combined phase: term by term on the wave-per-node kernel)
    ops.apply_M64(X, refM)
    ops.combined_k64(True)
    try:
        outK, outM = torch.empty_like(X), torch.empty_like(X)
        ops.apply_K64(X, outK)
        ops.apply_M64(X, outM)
        assert ops._k64grp is not None and ops._m64grp is not None
    finally:
        ops.combined_k64(False)
    assert ops._k64grp is None
    assert float((outK - refK).abs().max()) < 1e-12 * float(refK.abs().max())
    assert float((outM - refM).abs().max()) < 1e-13 * float(refM.abs().max())


def test_mfma_entry_point_refuses_what_it_does_not_serve(dev):
    """ds_spmm_union16m validates on the host before any launch: group size, table limits, aliasing, alignment."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(4)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    sysd = TetSystem(tm.vertices, tm.tets, 2, 2700.0)
    ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(8, 0))
    mt, L, p = ops._mfma, _hip.lib(), _hip.ptr
    X = torch.randn(sysd.n, 80, device=dev).bfloat16()
    Y, R0 = torch.empty_like(X), torch.empty_like(X)

    def call(G=8, max_entries=None, mbb=None, x=X, y=Y, ncols=80, epi=1, dinv=ops.dinv, y32=0):
        return L.ds_spmm_union16m(epi, G, 0, p(mt["gptr"]), p(mt["gcol"]), p(mt["gmeta"]), p(mt["gbase"]), p(mt["ghead"]), p(ops.kc), sysd.nnzb,
                                  (sysd.nv + G - 1) // G, mt["max_entries"] if max_entries is None else max_entries,
                                  mt["max_batch_blocks"] if mbb is None else mbb, sysd.nv, p(x), x.stride(0), p(y), y.stride(0), y32,
                                  p(R0), 80, None if dinv is None else p(dinv), ncols, 0.3, 0.7, 0, None, 0, _hip.stream_ptr())

    assert call() == 0
    for bad in (dict(G=4), dict(G=16), dict(max_entries=257), dict(mbb=0), dict(mbb=129), dict(y=X), dict(ncols=88), dict(ncols=78),
                dict(epi=0), dict(epi=3), dict(dinv=None), dict(y32=1, epi=2), dict(x=X[:, 1:])):
        assert call(**bad) != 0, bad
        assert L.ds_last_error()
    with pytest.raises(ValueError, match="mfma_groups"):
        HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(4, 0))


def test_mfma_terms_beside_other_kernels_are_repeatable(dev):
    """The MFMA term on one stream and the VALU kernels (the eigensolver's fp32 K X, the bf16 term of the corner-node
    level) on another, as the hypothesis lanes run them: every result of either stream equals its solo result bit for
    bit.  With v_mfma_f32_16x16x32_bf16 in the MFMA kernel AND packed FMAs in the VALU kernels this failed - hence the
    16x16x16 instructions of the one (csrc/spmm_mfma.inc) and the plain v_fma_f32 of the others (csrc/spmm_union.inc);
    test_immunity_to_foreign_mfma_and_the_probe_itself below isolates the hardware interaction."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    def make(cells, order, G, seed):
        v, t = meshgen.kuhn_box(cells)
        tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
        sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
        ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(max(G, 0), 0))
        g = torch.Generator(device=dev).manual_seed(seed)
        mk = lambda: torch.randn(sysd.n, 80, generator=g, device=dev).bfloat16()
        return dict(ops=ops, X=mk(), W=mk(), R=mk(), G=G)

    def term(c, out):
        if c["G"] < 0:
            c["ops"].apply_K(c["X32"], out)
            return
        out.copy_(c["W"])
        c["ops"].cheb_spmm16(c["X"], out, c["R"], 0.3, 0.7, False)

    mf = make(14, 2, 8, 1)
    assert mf["ops"]._mfma is not None and mf["ops"].kc is not None
    for other in (make(14, 2, -1, 2), make(16, 1, 0, 3)):
        if other["G"] < 0:
            other["X32"], other["W"] = other["X"].float(), other["W"].float()
        pair = (mf, other)
        for c in pair:
            c["ref"] = torch.empty_like(c["W"])
            term(c, c["ref"])
            c["outs"] = [torch.empty_like(c["W"]) for _ in range(16)]
            c["stream"] = torch.cuda.Stream()
        torch.cuda.synchronize()
        for _ in range(3):
            for c in pair:
                with torch.cuda.stream(c["stream"]):
                    for o in c["outs"]:
                        term(c, o)
            torch.cuda.synchronize()
            for c in pair:
                assert all(torch.equal(o, c["ref"]) for o in c["outs"]), c["G"]


def test_immunity_to_foreign_mfma_and_the_probe_itself(dev):
    """The gfx950 finding behind the library's "no packed FP32" build (csrc/Makefile): a REGISTER-ONLY spin of
    v_mfma_f32_16x16x32_bf16 (tests/probes/mfma_probe.hip: no memory, no LDS) on one stream changes results of a
    REGISTER-ONLY chain of v_pk_fma_f32 on another, never of the same chain of v_fma_f32, and the 16x16x16 / fp32 MFMA
    forms change nothing.  Asserted here: (a) the forms this library issues (bf16 16x16x16, fp32 16x16x4) leave both
    chains bit-exact; (b) the library's own accumulating kernels - fp32 K X, the bf16 VALU term, the MFMA term - return
    their solo results bit for bit beside the double-rate MFMA spin, the aggressor that corrupted 120 of 120 launches of
    the packed-FMA build (profiles/r03_mfma_interference_matrix.txt).  How often the packed chain is hit by the
    double-rate form is printed, not asserted (it is a property of the silicon, not of this library)."""
    import ctypes
    import os

    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libmfma_probe.so")
    if not os.path.exists(so):
        pytest.skip("tests/probes/libmfma_probe.so not built (make -C diffsound_amd/csrc probe)")
    P = ctypes.CDLL(so)
    P.probe_mfma_spin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    P.probe_fma_chain.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    spin_out = torch.empty(ncu * 4 * 64, device=dev)

    def beside(form, victim, outs, ref, iters=150000):
        assert P.probe_mfma_spin(form, iters, ncu * 4, spin_out.data_ptr(), sB.cuda_stream) == 0  # ~10 ms of MFMAs
        with torch.cuda.stream(sA):
            for o in outs:
                victim(o)
        torch.cuda.synchronize()
        return sum(0 if torch.equal(o, ref) else 1 for o in outs)

    seed = torch.rand(4096, device=dev) - 0.5
    hits = {}
    for packed in (1, 0):
        ref = torch.empty(ncu * 4 * 256 * 12, device=dev)

        def chain(o, packed=packed):
            assert P.probe_fma_chain(packed, 3000, ncu * 4, seed.data_ptr(), o.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream) == 0

        chain(ref)
        torch.cuda.synchronize()
        assert torch.isfinite(ref).all()
        outs = [torch.empty_like(ref) for _ in range(30)]
        for form in (16, 4):
            assert beside(form, chain, outs, ref) == 0, (packed, form)
        hits[packed] = sum(beside(32, chain, outs, ref) for _ in range(3))
    print(f"register-only chains beside the double-rate bf16 MFMA spin: v_pk_fma_f32 {hits[1]} of 90 launches changed, "
          f"v_fma_f32 {hits[0]} of 90")
    assert hits[0] == 0

    def make(cells, order, G, sd):
        v, t = meshgen.kuhn_box(cells)
        tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
        sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
        ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(max(G, 0), 0))
        g = torch.Generator(device=dev).manual_seed(sd)
        mk = lambda: torch.randn(sysd.n, 80, generator=g, device=dev).bfloat16()
        c = dict(ops=ops, X=mk(), W=mk(), R=mk(), G=G)
        if G < 0:
            c["X32"], c["W"] = c["X"].float(), c["W"].float()
        return c

    def term(c, out):
        if c["G"] < 0:
            c["ops"].apply_K(c["X32"], out)
            return
        out.copy_(c["W"])
        c["ops"].cheb_spmm16(c["X"], out, c["R"], 0.3, 0.7, False)

    for cells, order, G in ((16, 2, -1), (18, 1, 0), (16, 2, 8)):
        c = make(cells, order, G, 11)
        ref = torch.empty_like(c["W"])
        term(c, ref)
        torch.cuda.synchronize()
        outs = [torch.empty_like(ref) for _ in range(30)]
        for form in (32, 16):
            assert beside(form, lambda o, c=c: term(c, o), outs, ref) == 0, (G, form)


def test_fused_polish_products_match_separate_launches(dev):
    """ds_spmm_f64_polish (K_lambda X, K_mu X, M_s X in one walk) against three ds_spmm_bsr3 launches: bit-identical."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(7)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    sysd = TetSystem(tm.vertices, tm.tets, 2, 2700.0)
    ops = HipModalOps(sysd, 2e10, 3e10, two_level=False)
    L, p = _hip.lib(), _hip.ptr
    for c in (64, 84, 4):
        Xw = torch.randn((sysd.n, c + 8), generator=torch.Generator(device=dev).manual_seed(c), device=dev)
        X = Xw[:, 4:4 + c]  # a column range of a wider block
        ref = [torch.empty((sysd.n, c), dtype=torch.float64, device=dev) for _ in range(3)]
        for (kind, vals), out in zip(((2, sysd.klam), (2, sysd.kmu), (3, sysd.ms)), ref):
            ops._spmm(kind, vals, X, out)
        got = torch.full((3, sysd.n, c), float("nan"), dtype=torch.float64, device=dev)
        _hip.check(L.ds_spmm_f64_polish(p(sysd.rowptr), p(sysd.colidx), p(sysd.klam), p(sysd.kmu), p(sysd.ms), sysd.nv, p(X),
                                        X.stride(0), p(got[0]), p(got[1]), p(got[2]), c, c, _hip.stream_ptr()), "ds_spmm_f64_polish")
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
        # round 5: the same sums stored as fp32 blocks - exactly the fp64 results rounded once
        got32 = torch.full((3, sysd.n, c), float("nan"), dtype=torch.float32, device=dev)
        _hip.check(L.ds_spmm_f64_polish_f32out(p(sysd.rowptr), p(sysd.colidx), p(sysd.klam), p(sysd.kmu), p(sysd.ms), sysd.nv, p(X),
                                               X.stride(0), p(got32[0]), p(got32[1]), p(got32[2]), c, c, _hip.stream_ptr()),
                   "ds_spmm_f64_polish_f32out")
        for a, b in zip(got32, ref):
            assert torch.equal(a, b.float())
    # and through the ops: with fp64 blocks (the default) the read-out's Gram matrices are bit for bit what the separate launches
    # gave; with the optional fp32 blocks (fp32 matrix-core Gram behind them) they agree to ~1e-9 of |X_i| |Y_j|
    X = torch.randn((sysd.n, 64), generator=torch.Generator(device=dev).manual_seed(1), device=dev)
    Y = torch.empty((sysd.n, 64), dtype=torch.float64, device=dev)
    for f64 in (True, False):
        ops.polish_f32_blocks = not f64
        GK, coef, GM = ops.polish_products(X)
        for (kind, vals), G in zip(((2, sysd.klam), (2, sysd.kmu), (3, sysd.ms)), GK + [GM]):
            ops._spmm(kind, vals, X, Y)
            ref_g = ops.gram(X, Y)
            if f64:
                assert torch.equal(G, ref_g)
            else:
                scale = torch.sqrt(torch.outer((X.double() ** 2).sum(0), (Y ** 2).sum(0)))
                assert float(((G - ref_g).abs() / scale).max()) < 2e-8  # (few rows here: the 48-row folds average out at size)
        assert coef == list(ops.lame)
    ops.polish_f32_blocks = False


def test_polish_products(case, dev):
    h, c = case["hops"], case["cops"]
    g = torch.Generator().manual_seed(9)
    X = torch.randn((h.n, 16), generator=g)
    gk, coef, gm = h.polish_products(X.to(dev))
    wk, wcoef, wm = c.polish_products(X)
    assert np.allclose(coef, wcoef)
    for a, b in zip(gk + [gm], wk + [wm]):
        assert rel(a.cpu().numpy(), b.numpy()) < 2e-6  # assembly tolerance; the products themselves are fp64


@pytest.mark.parametrize("ncols", [4, 16, 64, 84, 86, 128])
def test_f64_value_spmm(case, dev, ncols):
    """fp64-value products (kinds 2 / 3: K_lambda, K_mu blocks and the mass scalars in fp64, fp32 X, fp64 result) against
    a dense fp64 product with the HIP-assembled values; <= 84 columns take the wave-per-node kernel, wider or
    odd-multiple blocks the generic one; strided views."""
    h = case["hops"]
    s_ = h.sys
    g = torch.Generator().manual_seed(ncols)
    big = torch.randn((h.n, ncols + 8), generator=g).to(dev)
    X = big[:, 4:4 + ncols]
    rp, ci = s_.rowptr.cpu().numpy(), s_.colidx.cpu().numpy()
    rows = np.repeat(np.arange(s_.nv), np.diff(rp))
    Xd = X.double().cpu().numpy()
    for kind, vals in ((2, s_.klam), (2, s_.kmu), (3, s_.ms)):
        Y = torch.full((h.n, ncols + 2), float("nan"), dtype=torch.float64, device=dev)[:, :ncols]
        h._spmm(kind, vals, X, Y)
        blocks = vals.cpu().numpy().reshape(-1, 3, 3) if kind == 2 else vals.cpu().numpy()[:, None, None] * np.eye(3)
        A = sp.bsr_matrix((blocks, ci, rp), shape=(h.n, h.n))
        ref = A @ Xd
        assert rel(Y.cpu().numpy(), ref) < 1e-13
        assert rows.shape[0] == ci.shape[0]


def test_rigid_basis(case):
    h = case["hops"]
    Y = h.rigid[:, :6].double().cpu().numpy()
    G = Y.T @ (case["M3"] @ Y)
    assert np.abs(G - np.eye(6)).max() < 1e-5
    K = case["lam"] * case["Kl"] + case["mu"] * case["Km"]
    assert np.abs(K @ Y).max() / (abs(K).max() * np.abs(Y).max()) < 1e-5


# ------------------------------------------------------------------------------------- oscillator
@pytest.mark.parametrize("A,m,F,S", [(1, 32, 150, 8000), (3, 16, 150, 4000), (2, 7, 1, 1500), (1, 64, 150, 8000)])
def test_oscillator_kernels(dev, A, m, F, S):
    from diffsound_amd import _hip

    L = _hip.lib()
    g = torch.Generator().manual_seed(A * 100 + m)
    f = torch.sort(torch.rand(m, generator=g) * 9000 + 300)[0].double()
    alpha, beta = 6.0, 1e-7
    w0 = 2 * np.pi * f
    d = 0.5 * (alpha + beta * w0 ** 2)
    w = torch.sqrt(w0 ** 2 - d ** 2)
    amp = torch.rand((A, m), generator=g) + 0.5
    force = torch.randn((A, F), generator=g)
    y = torch.empty((A, S), device=dev)
    p = _hip.ptr
    dd, wd, ad, fd = d.to(dev), w.to(dev), amp.to(dev), force.to(dev)
    _hip.check(L.ds_osc_bank_fwd(p(dd), p(wd), p(ad), p(fd), A, m, F, S, 32000.0, p(y), _hip.stream_ptr()), "fwd")
    ref = oosc.bank_closed_form_f64(f.numpy(), force.numpy(), S, 32000, alpha, beta, amp=amp.numpy())
    err = np.linalg.norm(y.cpu().numpy() - ref) / np.linalg.norm(ref)
    assert err < 1e-6, err
    # backward against fp64 autograd of the closed form
    gy = torch.randn((A, S), generator=g)
    dt_, wt_, at_ = d.clone().requires_grad_(True), w.clone().requires_grad_(True), amp.double().clone().requires_grad_(True)
    tau = (torch.arange(S, dtype=torch.float64) + 1) / 32000
    modes = torch.exp(-dt_[:, None] * tau[None]) * torch.sin(wt_[:, None] * tau[None])
    s = (at_[:, :, None] * modes[None]).sum(1)
    yy = torch.nn.functional.conv1d(s[None], torch.flip(force.double(), [-1])[:, None, :], groups=A, padding=F - 1)[0][:, :S]
    (yy * gy.double()).sum().backward()
    gs = torch.empty((A, S), device=dev)
    gd = torch.empty(m, dtype=torch.float64, device=dev)
    gw = torch.empty(m, dtype=torch.float64, device=dev)
    gamp = torch.empty((A, m), device=dev)
    _hip.check(L.ds_osc_bank_bwd(p(gy.to(dev)), p(dd), p(wd), p(ad), p(fd), A, m, F, S, 32000.0, p(gs), p(gd), p(gw),
                                 p(gamp), _hip.stream_ptr()), "bwd")
    assert rel(gd.cpu().numpy(), dt_.grad.numpy()) < 1e-5
    assert rel(gw.cpu().numpy(), wt_.grad.numpy()) < 1e-5
    assert rel(gamp.cpu().numpy(), at_.grad.numpy()) < 1e-5


def test_errors_are_loud(dev):
    from diffsound_amd import _hip

    L = _hip.lib()
    with pytest.raises(RuntimeError):
        _hip.check(L.ds_spmm_bsr3(7, None, None, None, None, 0, None, 0, None, 0, 0, None), "ds_spmm_bsr3")
    with pytest.raises(RuntimeError):
        from diffsound_amd.modal_ops import TetSystem

        TetSystem(torch.zeros((4, 3)), torch.zeros((1, 4), dtype=torch.int64), 1, 1000.0)  # CPU tensors


@pytest.mark.parametrize("mesh,order,ncols", [("2", 1, 8), ("3", 1, 8), ("bowl", 1, 40), ("bowl", 2, 24), ("6", 2, 72),
                                              ("6", 2, 80), ("6", 2, 84), ("5", 2, 4), ("3+unused", 1, 16)])
def test_union_spmm_matches_wave_per_node(dev, mesh, order, ncols):
    """ds_spmm_union (one wavefront per 4 nodes, shared neighbour panels gathered once; the default for <= 84
    columns) against the wave-per-node kernels on the same operands: K X, both Chebyshev-term epilogues and the
    residual epilogue, on strided views.  Repeated: the bugs met while building this kernel were intermittent
    (a stale descriptor word read 5 wait states too early).  The oracle comparisons of this file (test_spmm,
    test_fused_chebyshev_spmm, test_residual_and_cheb, the two-level tests) run through the same kernel."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    if mesh == "bowl":
        m = np.load("tests/golden/g0_bowl_mesh.npz")
        v, t = m["verts"], m["tets"]
    elif mesh == "3+unused":  # 9 clustered vertices that no tet references: empty rows and one whole empty group
        v, t = meshgen.kuhn_box(3)
        v = np.concatenate([v, np.array([[9.0, 9.0, 9.0 + 0.01 * i] for i in range(9)], dtype=v.dtype)], 0)
    else:
        v, t = meshgen.kuhn_box(int(mesh))
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
    ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
    assert sysd.groups is not None and sysd.groups["union"] is not None
    if mesh == "3+unused":
        assert int((sysd.rowptr[1:] == sysd.rowptr[:-1]).sum()) == 9
    u = sysd.groups["union"]
    ct, ut = u["ctab"].cpu().numpy(), u["utab"].cpu().numpy()
    assert ut.shape == ((sysd.nv + 3) // 4, 2) and ut[0, 0] == 0 and ut[-1, 1] == ct.shape[0]
    assert (ct[:, 3] - ct[:, 2]).max() <= u["capb"] and ct[-1, 3] == sysd.nnzb
    g = torch.Generator(device=dev).manual_seed(ncols)
    big = torch.randn((sysd.n, ncols + 16), generator=g, device=dev)
    X = big[:, 8:8 + ncols]
    Wp = torch.randn((sysd.n, ncols), generator=g, device=dev)
    R0 = torch.randn((sysd.n, ncols), generator=g, device=dev) * 1e10

    def run():
        Y = torch.zeros((sysd.n, ncols), device=dev)
        ops.apply_K(X, Y)
        a = Wp.clone()
        ops.cheb_spmm(X, a, R0, 0.31, 0.77, False)
        b = Wp.clone()
        ops.cheb_spmm(X, b, R0, 0.0, 0.5, True)
        c = torch.zeros((sysd.n, ncols), device=dev)
        ops.spmm_residual(X, R0, c)
        d = torch.full((sysd.n, ncols), float("nan"), device=dev)
        ops.apply_M(X, d)  # node-scalar values (epilogue 3)
        return Y, a, b, c, d

    assert ops._union_ok(X, Wp, R0) and ops.mgrp is not None
    # out-of-place form of the fused term: W_prev read from one block, the result written to a column range of another
    wide = torch.full((sysd.n, ncols + 8), float("nan"), device=dev)
    ops._union(1, X, wide[:, 4:4 + ncols], R0, 0.31, 0.77, False, Wprev=Wp)
    inplace = Wp.clone()
    ops._union(1, X, inplace, R0, 0.31, 0.77, False)
    assert torch.equal(wide[:, 4:4 + ncols], inplace)
    assert bool(torch.isnan(wide[:, :4]).all()) and bool(torch.isnan(wide[:, 4 + ncols:]).all())
    for _ in range(3):
        got = run()
        sysd.groups["union"] = None
        assert not ops._union_ok(X, Wp, R0)
        ref = run()
        sysd.groups["union"] = u
        for x, y in zip(got, ref):
            assert rel(x.cpu().numpy(), y.cpu().numpy()) < 5e-6


def test_union_spmm_operands_beyond_2gb(dev):
    """Operand blocks of >= 2 GB (configs[4]: n = 4.1 M rows of a 416-column basis buffer) take the per-panel
    descriptor variant of ds_spmm_union.  Here a small mesh with a huge leading dimension: every operand is an
    80-column range of a (n x 81 000)-float buffer, 2.1 GB from first to last row; all four epilogues against the
    same products on compact blocks (bitwise: the arithmetic is the same, only the addressing differs)."""
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    v, t = meshgen.kuhn_box(6)
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    sysd = TetSystem(tm.vertices, tm.tets, 2, 2700.0)
    ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
    ncols, ld = 80, 81000
    assert 3 * sysd.nv * ld * 4 >= 0x7F000000
    g = torch.Generator(device=dev).manual_seed(7)
    small = [torch.randn((sysd.n, ncols), generator=g, device=dev) for _ in range(3)]
    small[2] *= 1e10
    wide = [torch.empty((sysd.n, ld), device=dev) for _ in range(4)]  # X, W_prev, R0, out: 2.1 GB each
    Xb, Wpb, R0b = (w[:, 16:16 + ncols] for w in wide[:3])
    for dst, src in zip((Xb, Wpb, R0b), small):
        dst.copy_(src)
    Xs, Wps, R0s = small
    assert ops._union_ok(Xb, Wpb, R0b)

    def run(X, Wp, R0, out_of):
        Y = out_of()
        ops.apply_K(X, Y)
        a = out_of()
        a.copy_(Wp)
        ops.cheb_spmm(X, a, R0, 0.31, 0.77, False)
        c = out_of()
        ops.spmm_residual(X, R0, c)
        d = out_of()
        ops.apply_M(X, d)
        e = out_of()
        ops._union(1, X, e, R0, 0.31, 0.77, False, Wprev=Wp)  # out-of-place form
        return [z.clone() for z in (Y, a, c, d, e)]

    ref = run(Xs, Wps, R0s, lambda: torch.zeros((sysd.n, ncols), device=dev))
    outw = wide[3]
    got = run(Xb, Wpb, R0b, lambda: outw[:, 32:32 + ncols])

    def one(fn_big, fn_small):  # one product at a time: wide operands against compact ones
        o_b = outw[:, 32:32 + ncols]
        o_s = torch.zeros((sysd.n, ncols), device=dev)
        fn_big(o_b)
        fn_small(o_s)
        assert torch.equal(o_b, o_s)

    one(lambda o: ops.apply_K(Xb, o), lambda o: ops.apply_K(Xs, o))
    one(lambda o: ops.spmm_residual(Xb, R0b, o), lambda o: ops.spmm_residual(Xs, R0s, o))
    one(lambda o: ops.apply_M(Xb, o), lambda o: ops.apply_M(Xs, o))
    one(lambda o: ops._union(1, Xb, o, R0b, 0.31, 0.77, False, Wprev=Wpb),
        lambda o: ops._union(1, Xs, o, R0s, 0.31, 0.77, False, Wprev=Wps))
    one(lambda o: (o.copy_(Wpb), ops.cheb_spmm(Xb, o, R0b, 0.31, 0.77, False)),
        lambda o: (o.copy_(Wps), ops.cheb_spmm(Xs, o, R0s, 0.31, 0.77, False)))
    # round 5: the fused residual (epilogue 4) and [K X | M X] in one walk (epilogue 5) on operands beyond 2 GB as well
    lam = (torch.rand(ncols, generator=g, device=dev, dtype=torch.float64) + 0.5) * 1e9
    norms = []

    def resid(X):
        def go(o):
            norms.append(ops.residual_fused(X, lam, o))
        return go

    assert ops.residual_fused_ok(Xb, outw[:, 32:32 + ncols])
    one(resid(Xb), resid(Xs))
    assert torch.equal(norms[0][0], norms[1][0]) and torch.equal(norms[0][1], norms[1][1])
    mw, ms_ = wide[1][:, 200:200 + ncols], torch.zeros((sysd.n, ncols), device=dev)
    one(lambda o: ops.apply_KM(Xb, o, mw), lambda o: ops.apply_KM(Xs, o, ms_))
    assert torch.equal(mw, ms_)
    assert all(torch.isfinite(r).all() for r in ref + got)


@pytest.mark.parametrize("mesh,order", [("2", 1), ("3", 2), ("bowl", 1), ("bowl", 2), ("6", 2), ("3+unused", 1)])
def test_device_symbolic_phase_matches_host(dev, mesh, order):
    """ds_dpattern_build (pattern, contribution lists, neighbour-union tables, chunk table - rocPRIM sorts + small
    kernels on the device) against the host routines ds_pattern_build / ds_groups_build / union_chunks: every array
    identical; and against the reference-generated goldens through test_pattern_and_assembly (nnz of the bowl)."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import UNION_CAP

    if mesh == "bowl":
        m = np.load("tests/golden/g0_bowl_mesh.npz")
        v, t = m["verts"], m["tets"]
    elif mesh == "3+unused":
        v, t = meshgen.kuhn_box(3)
        v = np.concatenate([v, np.array([[9.0, 9.0, 9.0 + 0.01 * i] for i in range(9)], dtype=v.dtype)], 0)
    else:
        v, t = meshgen.kuhn_box(int(mesh))
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    nv = tm.vertices.shape[0]
    tets = tm.tets.to(torch.int32).contiguous()
    d = _hip.DevicePattern(tets, nv, UNION_CAP)
    h = _hip.Pattern(tets.cpu(), nv)
    assert (d.nnzb, d.ncontrib) == (h.nnzb, h.ncontrib)
    for name in ("rowptr", "colidx", "diagidx", "cptr", "clist"):
        assert torch.equal(getattr(d, name).cpu(), getattr(h, name)), name
    g = _hip.Groups(h.rowptr, h.colidx, nv)
    assert (d.ngroups, d.ne) == (g.ngroups, g.ne)
    for name in ("gptr", "gent", "goff", "kperm"):
        assert torch.equal(getattr(d, name).cpu(), getattr(g, name)), name
    ut, ct = _hip.union_chunks(g.gptr, g.goff, UNION_CAP)
    assert d.single == (ct.shape[0] == ut.shape[0])
    assert torch.equal(d.utab.cpu(), ut) and torch.equal(d.ctab.cpu(), ct)
    if mesh == "bowl":  # the reference's coalesced K has exactly these many scalar non-zeros
        assert d.nnzb * 9 == {1: 294381, 2: 3674016}[order]
    # groups larger than the cap are cut into several chunks, by the same greedy rule as on the host
    for cap in (8, 20):
        small = _hip.DevicePattern(tets, nv, cap)
        ut, ct = _hip.union_chunks(g.gptr, g.goff, cap)
        assert not small.single and small.nchunks == ct.shape[0]
        assert torch.equal(small.utab.cpu(), ut) and torch.equal(small.ctab.cpu(), ct)
        assert int((small.ctab[:, 1] - small.ctab[:, 0]).max()) <= cap and int((small.ctab[:, 3] - small.ctab[:, 2]).max()) <= cap


def test_mesh_front_end_primitives(dev):
    """ds_unique_rows3 against torch.unique(dim=0, return_inverse=True) on the host (duplicates, negative values,
    -0.0 / +0.0, denormals) and ds_edge_table against a NumPy edge set; then the lifted bowl against the reference's
    own output for it (G3 fixture: o2_vertices / o2_tets, bit-identical) and the torch path."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh

    rng = np.random.default_rng(5)
    base = rng.standard_normal((3000, 3)).astype(np.float32)
    base[:50, 0] = 0.0
    base[50:100, 0] = -0.0
    base[100:120] = np.float32(1e-42)  # denormal rows
    base[120:200, 1] = base[120:200, 0]
    x = np.concatenate([base, base[rng.integers(0, 3000, size=5000)], -base[:500]])
    x = x[rng.permutation(len(x))]
    xt = torch.from_numpy(x)
    uniq, inv_ref = torch.unique(xt, dim=0, return_inverse=True)
    inv, first = _hip.unique_rows3(xt.to(dev))
    assert first.shape[0] == uniq.shape[0]
    assert torch.equal(inv.cpu(), inv_ref)
    assert torch.equal(xt[first.cpu()].abs(), uniq.abs()) and torch.equal(xt[first.cpu()] == 0, uniq == 0)
    # representative = lowest original index of the group
    low = torch.full((uniq.shape[0],), len(x), dtype=torch.long).scatter_reduce_(0, inv_ref, torch.arange(len(x)), reduce="amin")
    assert torch.equal(first.cpu(), low)
    # edge table
    v, t = meshgen.kuhn_box(5)
    t = t.astype(np.int64)
    ea, eb, te = _hip.edge_table(torch.from_numpy(t).to(dev), len(v))
    pairs = np.array([(0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)])
    e = np.sort(t[:, pairs], axis=2)  # (T, 6, 2)
    uniq_e, inv_e = np.unique(e.reshape(-1, 2), axis=0, return_inverse=True)
    assert np.array_equal(np.stack([ea.cpu().numpy(), eb.cpu().numpy()], 1), uniq_e)
    assert np.array_equal(te.cpu().numpy(), inv_e.reshape(-1, 6))
    with pytest.raises(RuntimeError, match="outside"):
        _hip.edge_table(torch.from_numpy(t).to(dev), len(v) - 1)
    # lifted bowl: device path (edge table + radix unique) == torch path on the host, bit for bit
    m = np.load("tests/golden/g0_bowl_mesh.npz")
    g = np.load("tests/golden/g3_bowl_o2.npz")
    vt, tt = torch.from_numpy(m["verts"]), torch.from_numpy(m["tets"]).long()
    d = TetMesh(vt.to(dev), tt.to(dev)).to_high_order(2)
    h = TetMesh(vt, tt).to_high_order(2)
    assert torch.equal(d.vertices.cpu(), h.vertices) and torch.equal(d.tets.cpu(), h.tets)
    assert np.array_equal(d.vertices.cpu().numpy(), g["o2_vertices"]) and np.array_equal(d.tets.cpu().numpy(), g["o2_tets"])
    # midpoints stay differentiable w.r.t. the corners, same gradient as the host path
    vd = vt.to(dev).requires_grad_(True)
    md = TetMesh(vd, tt.to(dev)).to_high_order(2)
    w = torch.randn(md.vertices.shape, generator=torch.Generator().manual_seed(1))
    (md.vertices * w.to(dev)).sum().backward()
    vh = vt.clone().requires_grad_(True)
    (TetMesh(vh, tt).to_high_order(2).vertices * w).sum().backward()
    assert torch.allclose(vd.grad.cpu(), vh.grad, rtol=1e-5, atol=1e-6)
