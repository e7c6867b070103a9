"""N > 1 path on CPU: two gloo ranks shard the material hypotheses round-robin, run the solver on
their shard (oracle ops standing in for the HIP kernels) and all-reduce the scalar loss - the only
collective of the data path."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _hyp_loss(E, nu):
    from diffsound_amd import meshgen
    from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
    from oracle import fem
    from oracle.ops_cpu import CpuModalOps

    v, t = meshgen.kuhn_box(3)
    v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), 1)
    d = fem.OracleDeform(v, t, 1)
    Kl, Km = fem.assemble_stiffness(d, 1.0, 0.0), fem.assemble_stiffness(d, 0.0, 1.0)
    M3, _ = fem.assemble_mass(v, t, 1, 2700.0)
    lam, mu = fem.lame(E, nu)
    ops = CpuModalOps(Kl, Km, M3, v.numpy(), lam, mu, dtype=torch.float64)
    res = ModalSolver(ops, SolverConfig(block=8, cheb_degree=4)).solve(6)
    f = torch.sqrt(res.eigenvalues) / (2 * np.pi)
    return float(((f / 1e4) ** 2).mean())


def _worker(rank, world, port, hyps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffsound_amd.pipeline import all_reduce_loss, shard_hypotheses

    torch.set_num_threads(2)
    mine = shard_hypotheses(len(hyps), rank, world)
    local = sum(_hyp_loss(*hyps[i]) for i in mine)
    total = all_reduce_loss(local, torch.device("cpu"))
    out[rank] = (mine, local, total)
    dist.destroy_process_group()


def test_two_rank_hypothesis_sharding():
    hyps = [(2e10 + 1e10 * i, 0.15 + 0.05 * i) for i in range(4)]
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), hyps, out), nprocs=world, join=True)
    serial = sum(_hyp_loss(*h) for h in hyps)
    assert sorted(out[0][0] + out[1][0]) == list(range(len(hyps)))
    assert abs(out[0][2] - serial) < 1e-12 * max(1.0, abs(serial))
    assert abs(out[1][2] - out[0][2]) == 0.0
    assert abs(out[0][1] + out[1][1] - serial) < 1e-12 * max(1.0, abs(serial))


def _worker8(rank, world, port, nhyp, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffsound_amd.pipeline import all_reduce_loss, gather_rank_stats, shard_hypotheses

    torch.set_num_threads(1)
    rng = np.random.default_rng(2024)  # the hypothesis draw of bench.py
    Es, nus = rng.uniform(1e10, 1e11, size=64), rng.uniform(0.1, 0.4, size=64)
    mine = shard_hypotheses(nhyp, rank, world)
    local = sum(float(Es[h] * 1e-11 + nus[h]) for h in mine)  # stand-in for a pass's scalar loss
    total = all_reduce_loss(local, torch.device("cpu"))
    stats = gather_rank_stats([len(mine), sum(mine), rank], torch.device("cpu"))
    out[rank] = (mine, total, stats)
    dist.destroy_process_group()


def test_eight_rank_layout_of_configs3():
    """configs[3]: 64 hypotheses over 8 ranks - every rank owns 8 of them (none idle), the shards cover the batch
    exactly once, the all-reduced loss equals the serial sum on every rank, the per-rank statistics line up."""
    world, nhyp = 8, 64
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker8, args=(world, _free_port(), nhyp, out), nprocs=world, join=True)
    rng = np.random.default_rng(2024)
    Es, nus = rng.uniform(1e10, 1e11, size=64), rng.uniform(0.1, 0.4, size=64)
    serial = sum(float(Es[h] * 1e-11 + nus[h]) for h in range(nhyp))
    owned = sorted(h for r in range(world) for h in out[r][0])
    assert owned == list(range(nhyp)) and all(len(out[r][0]) == 8 for r in range(world))
    for r in range(world):
        assert abs(out[r][1] - serial) < 1e-12 * serial
        assert out[r][2] == out[0][2] and len(out[r][2]) == world
        assert out[r][2][r] == [8.0, float(sum(out[r][0])), float(r)]
