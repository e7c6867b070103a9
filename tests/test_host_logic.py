"""Host-side logic of the product package that needs no GPU: element tables vs the reference's
(golden) constants, synthetic mesh generator, Morton ordering, Gmsh I/O, hypothesis sharding."""
import numpy as np
import pytest
import torch

from diffsound_amd import fem_tables, meshgen
from diffsound_amd.diffelastic import mesh as dmesh
from diffsound_amd.modal_ops import morton_order
from diffsound_amd.pipeline import shard_hypotheses


@pytest.mark.parametrize("order", [1, 2])
def test_tables_match_reference_constants(golden, order):
    g = golden("g1_constants.npz")
    pts, w = fem_tables.gauss_rule(order)
    assert np.array_equal(pts, g[f"gauss_pts_o{order}"]) and np.array_equal(w, g[f"gauss_w_o{order}"])
    assert np.array_equal(fem_tables.shape_values(pts, order), g[f"N_o{order}"])
    assert np.array_equal(fem_tables.shape_gradients(pts, order), g[f"dN_dL_o{order}"])
    N = fem_tables.NODES_PER_TET[order]
    ref_mm = g[f"elem_mass_o{order}"].reshape(N, 3, N, 3)[:, 0, :, 0]
    assert np.array_equal(fem_tables.mass_table_f32(order), ref_mm)
    # stiffness table = the reference's Gauss sum of dN (x) dN
    dN = g[f"dN_dL_o{order}"].astype(np.float64)
    ref = np.einsum("g,gak,gbl->akbl", g[f"gauss_w_o{order}"].astype(np.float64), dN, dN)
    assert np.allclose(fem_tables.stiffness_table(order), ref, rtol=0, atol=1e-15)


def test_kuhn_box_matches_fixture(golden):
    g = golden("g2_cube2.npz")
    v, t = meshgen.kuhn_box(2)
    assert np.array_equal(v, g["verts"]) and np.array_equal(t, g["tets"])
    v, t = meshgen.kuhn_box(26)
    assert t.shape == (105456, 4) and v.shape == (27 ** 3, 3)
    # conforming, positive total volume = box volume
    p = v[t].astype(np.float64)
    vol = np.abs(np.linalg.det(p[:, :3] - p[:, 3:4])).sum() / 6
    assert abs(vol / (0.10 * 0.08 * 0.06) - 1) < 1e-5


def test_to_high_order_matches_reference(golden):
    g = golden("g2_cube2.npz")
    m = dmesh.TetMesh(torch.from_numpy(g["verts"]), torch.from_numpy(g["tets"])).to_high_order(2)
    assert np.array_equal(m.vertices.numpy(), g["o2_vertices"]) and np.array_equal(m.tets.numpy(), g["o2_tets"])
    assert np.array_equal(m.transform_matrix.numpy(), g["o2_transform"])


def test_morton_order_is_a_locality_preserving_permutation():
    v, _ = meshgen.kuhn_box(8)
    perm = morton_order(torch.from_numpy(v))
    assert sorted(perm.tolist()) == list(range(len(v)))
    p = v[perm.numpy()]
    # consecutive nodes are close: mean hop is a small multiple of the grid step
    hop = np.linalg.norm(np.diff(p, axis=0), axis=1).mean()
    assert hop < 3 * 0.1 / 8


def _union_fill(v, t, perm, G):
    """entries of the unions of G consecutive rows / blocks of the node-adjacency pattern under the ordering ``perm``"""
    nv, N = v.shape[0], t.shape[1]
    rows, cols = np.repeat(t, N, axis=1).reshape(-1), np.tile(t, (1, N)).reshape(-1)
    key = np.unique(rows.astype(np.int64) * nv + cols)
    inv = np.empty(nv, dtype=np.int64)
    inv[perm] = np.arange(nv)
    return np.unique((inv[key // nv] // G) * nv + inv[key % nv]).size / key.size


def test_node_ordering_aligns_bricks_with_the_planes_of_a_structured_mesh(golden):
    """The neighbour-union SpMM kernels walk the union of the rows of 4 / 8 consecutive nodes: the ordering decides how
    much of the rows' blocks that union is.  On the benchmark's kind of mesh (jittered Kuhn box, ord-2) bricks aligned to
    the node planes give <= 0.47 / 0.30 (a Morton curve over the raw coordinates: 0.58 / 0.43 at the benchmark size); an
    unjittered box and a plate are recognised as well; an unstructured mesh falls back to quantile slabs (no worse than
    before within a few per cent)."""
    from oracle import fem

    for cells, jitter, lim4, lim8 in ((8, 0.15, 0.47, 0.30), (6, 0.0, 0.47, 0.30)):
        v, t = meshgen.kuhn_box(cells, jitter=jitter)
        vo, to = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), 2)
        perm = morton_order(vo).numpy()
        assert sorted(perm.tolist()) == list(range(vo.shape[0]))
        assert _union_fill(vo.numpy(), to.numpy(), perm, 4) < lim4 and _union_fill(vo.numpy(), to.numpy(), perm, 8) < lim8
    v, t = meshgen.kuhn_box(20, 20, 1, box=(0.2, 0.2, 0.005))
    perm = morton_order(torch.from_numpy(v)).numpy()
    assert _union_fill(v, t.astype(np.int64), perm, 8) < 0.36
    m = golden("g0_bowl_mesh.npz")
    perm = morton_order(torch.from_numpy(m["verts"])).numpy()
    assert sorted(perm.tolist()) == list(range(m["verts"].shape[0]))
    assert _union_fill(m["verts"], m["tets"].astype(np.int64), perm, 8) < 0.48


def test_gmsh_roundtrip(tmp_path):
    v, t = meshgen.kuhn_box(2)
    path = str(tmp_path / "m.msh")
    dmesh.write_gmsh22(path, v, t)
    pts, tets = dmesh.read_gmsh22(path)
    assert np.allclose(pts, v) and np.array_equal(tets, t)


def test_gmsh_reader_on_reference_style_file(golden):
    # the bowl fixture was read from the reference's Gmsh 2.2 binary file; re-emit and re-read it
    m = golden("g0_bowl_mesh.npz")
    import tempfile, os

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "bowl.obj_.msh")
        dmesh.write_gmsh22(path, m["verts"], m["tets"])
        pts, tets = dmesh.read_gmsh22(path)
    assert np.array_equal(pts.astype(np.float32), m["verts"]) and np.array_equal(tets, m["tets"])


def test_gmsh_reader_and_writer_against_a_file_the_reference_ships(golden, tmp_path):
    """Byte-format pin: tests/golden/oloid.msh is data/mesh/shape/oloid.msh of the reference (Gmsh 2.2
    binary as fTetWild wrote it; copied by make_golden.py).  (1) read_gmsh22 returns the file's nodes and tetrahedra bit for
    bit; (2) TetMesh.import_from_file gives exactly what the REFERENCE's loader gave (src/diffelastic/mesh.py:181-199:
    float cast + remove_duplicate_vertices - lexicographic node order, lowest-index representative); (3) write_gmsh22 of what
    was read reproduces the file's $MeshFormat / $Nodes / $Elements sections byte for byte - up to the newline this writer (as
    Gmsh itself and meshio) puts behind each binary section (before $EndMeshFormat / $EndNodes / $EndElements), which fTetWild omits; the file's trailing $ElementData
    section (a per-element colour the reference never reads) is not mesh data."""
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oloid.msh")
    g = golden("g8_oloid_import.npz")
    pts, tets = dmesh.read_gmsh22(path)
    assert pts.dtype == np.float64 and np.array_equal(pts, g["raw_points"]) and np.array_equal(tets, g["raw_tets"])
    # (import_from_file itself puts the mesh on the HIP device - tests/test_parity_gpu.py runs it there; here its steps on
    # host tensors: float cast, then the duplicate merge)
    m = dmesh.TetMesh(torch.from_numpy(pts).float(), torch.from_numpy(tets[:, :4]).long())
    m.remove_duplicate_vertices()
    assert int(g["order"]) == 1 and m.vertices.dtype == torch.float32
    assert np.array_equal(m.vertices.numpy(), g["vertices"]) and np.array_equal(m.tets.numpy(), g["tets"])
    out = str(tmp_path / "oloid_out.msh")
    dmesh.write_gmsh22(out, pts, tets)
    a, b = open(path, "rb").read(), open(out, "rb").read()
    a = a[:a.index(b"$EndElements\n") + len(b"$EndElements\n")]
    for tag in (b"$EndMeshFormat", b"$EndNodes", b"$EndElements"):  # (the newline behind a binary section, see above)
        b = b.replace(b"\n" + tag, tag)
    assert b == a
    p2, t2 = dmesh.read_gmsh22(out)
    assert np.array_equal(p2, pts) and np.array_equal(t2, tets)


def test_shard_hypotheses_partitions():
    for num, world in ((64, 8), (7, 2), (3, 4)):
        parts = [shard_hypotheses(num, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(num))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_largest_connected_component_matches_scipy():
    """Device-agnostic label propagation vs the reference's scipy.sparse.csgraph route
    (reference src/dmtet/geometry/dmtet_thickness.py:254-285)."""
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csgraph

    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import largest_connected_component

    v1, t1 = meshgen.kuhn_box(3)
    v2, t2 = meshgen.kuhn_box(2)
    v = np.concatenate([v2 + 10.0, v1, v2 - 10.0])  # three bodies, the middle one is the largest
    t = np.concatenate([t2, t1 + len(v2), t2 + len(v2) + len(v1)])
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(v))  # scramble node numbering
    inv = np.empty_like(perm); inv[perm] = np.arange(len(v))
    v, t = v[perm], inv[t]
    vo, to = largest_connected_component(torch.from_numpy(v), torch.from_numpy(t))
    rows = np.concatenate([t[:, i] for i in range(4)]); cols = np.concatenate([t[:, (i + 1) % 4] for i in range(4)])
    A = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(len(v), len(v))).tocsr()
    _, labels = csgraph.connected_components(A, directed=False)
    big = np.argmax(np.bincount(labels))
    assert np.array_equal(vo.numpy(), v[labels == big])
    assert to.shape[0] == len(t1) and int(to.max()) == len(v1) - 1
    assert np.allclose(vo.numpy()[to.numpy()], v[t[(labels[t] == big).all(1)]])
    # a single body comes back untouched
    a, b = largest_connected_component(torch.from_numpy(v1), torch.from_numpy(t1))
    assert a.shape[0] == len(v1) and b.shape[0] == len(t1)


def test_mfma_table_builder_on_the_host():
    """The topology tables of the MFMA term kernel are plain tensor algebra (modal_ops.TetSystem._build_mfma_tables): run here
    on CPU tensors for a random symmetric pattern with groups of very different sizes - unions, presence masks, block offsets,
    the fixed-stride head records and the blocks of the fullest batch under the partition the kernel walks (with and without the
    tail batch).  The device run of the same builder is checked against the BSR pattern in tests/test_hip_kernels.py."""
    from types import SimpleNamespace

    from diffsound_amd.modal_ops import MF_BATCH, MF_TAIL, TetSystem

    rng = np.random.default_rng(3)
    for nv, deg, expect_tail in ((203, 9, True), (160, 70, False)):
        A = np.zeros((nv, nv), dtype=bool)
        for r in range(nv):
            A[r, rng.choice(nv, size=rng.integers(1, deg), replace=False)] = True
        A |= A.T
        A[np.arange(nv), np.arange(nv)] = True
        rowptr = np.concatenate([[0], np.cumsum(A.sum(1))]).astype(np.int32)
        colidx = np.concatenate([np.flatnonzero(A[r]) for r in range(nv)]).astype(np.int32)
        fake = SimpleNamespace(nv=nv, device=torch.device("cpu"), rowptr=torch.from_numpy(rowptr), colidx=torch.from_numpy(colidx))
        mt = TetSystem._build_mfma_tables(fake, 8, MF_BATCH)
        gptr, gcol, gmeta, ghead = (mt[k].numpy().astype(np.int64) for k in ("gptr", "gcol", "gmeta", "ghead"))
        ng = (nv + 7) // 8
        assert mt["ngroups"] == ng and gptr[-1] == gcol.size
        assert np.array_equal(np.sort(mt["kperm"].numpy()), np.arange(colidx.size))  # every block exactly once
        most = worst_tail = worst_plain = 0
        for g in range(ng):
            rows = np.arange(8 * g, min(8 * g + 8, nv))
            union = np.flatnonzero(A[rows].any(0))
            e0, e1 = gptr[g], gptr[g + 1]
            assert np.array_equal(gcol[e0:e1], union)
            mask = (A[rows][:, union] * (1 << np.arange(rows.size))[:, None]).sum(0)
            assert np.array_equal(gmeta[e0:e1] & 0xff, mask)
            per_entry = A[rows][:, union].sum(0)
            assert np.array_equal(gmeta[e0:e1] >> 8, np.concatenate([[0], np.cumsum(per_entry)[:-1]]))
            ne = union.size
            h = min(ne, 64)
            assert np.array_equal(ghead[g, :h], union[:h]) and not ghead[g, h:64].any()
            assert np.array_equal(ghead[g, 64:64 + h], gmeta[e0:e0 + h]) and not ghead[g, 64 + h:].any()
            most = max(most, ne)
            nb = max(1, (ne - MF_TAIL + MF_BATCH - 1) // MF_BATCH)
            for b in range(nb):
                hi = ne if b == nb - 1 else (b + 1) * MF_BATCH
                worst_tail = max(worst_tail, int(per_entry[b * MF_BATCH:hi].sum()))
            for b in range(0, ne, MF_BATCH):
                worst_plain = max(worst_plain, int(per_entry[b:b + MF_BATCH].sum()))
        assert mt["max_entries"] == most and (most <= 128) == expect_tail
        assert mt["max_batch_blocks"] == (worst_tail if expect_tail else worst_plain)


def _fake_sysfs(tmp_path, gpus):
    """A KFD topology with one CPU node and the given GPUs [(bus, numa_node, cpulist)] under tmp_path."""
    nodes = tmp_path / "sys/class/kfd/kfd/topology/nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (bus, numa, cpus) in enumerate(gpus):
        d = nodes / str(i + 1)
        d.mkdir()
        d.joinpath("properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
        p = tmp_path / f"sys/bus/pci/devices/0000:{bus:02x}:00.0"
        p.mkdir(parents=True)
        p.joinpath("numa_node").write_text(f"{numa}\n")
        p.joinpath("local_cpulist").write_text(cpus + "\n")
    return str(tmp_path)


def test_rank_cpu_affinity_from_sysfs(tmp_path):
    """diffsound_amd.hostcpu: a rank's CPUs = the NUMA node of ITS device, split among the ranks that share the node, from sysfs
    alone (no HIP call); unknown topologies leave the affinity untouched and say so."""
    import os

    from diffsound_amd import hostcpu

    assert hostcpu.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert hostcpu.compact([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    ncpu = len(os.sched_getaffinity(0))
    if ncpu < 8:
        pytest.skip("needs 8 allowed CPUs")
    half = ncpu // 2
    lo, hi = f"0-{half - 1}", f"{half}-{ncpu - 1}"
    root = _fake_sysfs(tmp_path, [(0x11, 0, lo), (0x21, 0, lo), (0x91, 1, hi), (0xa1, 1, hi)])
    assert hostcpu.kfd_gpus(root) == ["0000:11:00.0", "0000:21:00.0", "0000:91:00.0", "0000:a1:00.0"]
    env = {}
    recs = [hostcpu.bind_rank_to_device_numa(r, 4, min_cpus=2, root=root, env=env, apply=False) for r in range(4)]
    assert [r["numa_node"] for r in recs] == [0, 0, 1, 1]
    q = half // 2
    assert recs[0]["cpu_list"] == hostcpu.compact(range(0, q)) and recs[1]["cpu_list"] == hostcpu.compact(range(q, 2 * q))
    assert recs[2]["cpu_list"] == hostcpu.compact(range(half, half + (ncpu - half) // 2))
    assert all(r["cpus"] >= 2 and r["pci"] for r in recs)
    # a visible-device list re-maps the local ranks
    r = hostcpu.bind_rank_to_device_numa(0, 1, min_cpus=2, root=root, env={"HIP_VISIBLE_DEVICES": "2"}, apply=False)
    assert r["numa_node"] == 1 and r["pci"] == "0000:91:00.0"
    # no topology / a device without a NUMA node of its own: nothing bound, reason given
    r = hostcpu.bind_rank_to_device_numa(0, 1, root=str(tmp_path / "nowhere"), env=env, apply=False)
    assert not r["bound"] and "KFD" in r["why"]
    root2 = _fake_sysfs(tmp_path / "b", [(0x11, -1, f"0-{ncpu - 1}")])
    r = hostcpu.bind_rank_to_device_numa(0, 1, root=root2, env=env, apply=False)
    assert not r["bound"] and r["numa_node"] is None
    # and for real, in a child process so this test's own affinity stays as it is
    import subprocess
    import sys

    code = ("import os, json; from diffsound_amd import hostcpu; "
            f"r = hostcpu.bind_rank_to_device_numa(1, 4, min_cpus=2, root={root!r}, env={{}}); "
            "print(json.dumps([r['bound'], sorted(os.sched_getaffinity(0))]))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert out.returncode == 0, out.stderr
    import json

    bound, cpus = json.loads(out.stdout.strip().splitlines()[-1])
    assert bound and cpus == list(range(q, 2 * q))
