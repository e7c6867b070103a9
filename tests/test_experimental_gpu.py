"""Parked SpMM experiment (the batched kernel of diffsound_amd/csrc/spmm_experimental.inc): parity with the
wave-per-node kernels.
Only runs against a library built with `make -C diffsound_amd/csrc EXPERIMENTAL=1` (skipped otherwise)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from diffsound_amd import _hip

    if not hasattr(_hip.lib(), "ds_spmm_batched"):
        pytest.skip("library built without EXPERIMENTAL=1")
    return torch.device("cuda:0")


@pytest.mark.parametrize("mesh,order,ncols", [("3", 1, 8), ("bowl", 1, 40), ("6", 2, 72), ("6", 2, 80)])
def test_experimental_spmm_matches_production(dev, monkeypatch, mesh, order, ncols):
    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.modal_ops import HipModalOps, TetSystem

    monkeypatch.setenv("DS_SPMM_BATCHED", "1")
    monkeypatch.setenv("DS_SPMM_UNION", "0")  # the batched kernel is compared with the wave-per-node kernels
    if mesh == "bowl":
        m = np.load("tests/golden/g0_bowl_mesh.npz")
        v, t = m["verts"], m["tets"]
    else:
        v, t = meshgen.kuhn_box(int(mesh))
    tm = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(tm.vertices, tm.tets, order, 2700.0)
    ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
    assert ops.batches is not None
    g = torch.Generator(device=dev).manual_seed(ncols)
    big = torch.randn((sysd.n, ncols + 16), generator=g, device=dev)
    X = big[:, 8:8 + ncols]  # a strided view, as in the solver
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
        return Y, a, b, c

    for _ in range(3):  # the bugs found while building these kernels were intermittent
        got = run()
        saved = ops.batches
        ops.batches = None
        ref = run()
        ops.batches = saved
        for x, y in zip(got, ref):
            assert rel(x, y) < 5e-6
