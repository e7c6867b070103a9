"""The N > 1 path of bench.py on real devices: `python bench.py --gpus 2` must start two ranks by itself, shard the
hypotheses round-robin, all-reduce the scalar loss and report n_gpus = 2 with the same loss sum as one rank running
all hypotheses.  With >= 2 visible devices the collective is RCCL (backend "nccl"); on a one-GPU box the same
path is rehearsed with two gloo ranks sharing the device.   pytest -m gpu."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "1", "--warmup", "0", "--cells", "6", "--modes", "16", "--block", "24", "--no-cpu-baseline"]


def _bench(*args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, out.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(line[0])


@pytest.fixture(scope="module")
def ndev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.cuda.device_count()


@pytest.fixture(scope="module")
def one_rank(ndev):
    return _bench("--gpus", "1", "--hyp-per-gpu", "4", "--lanes", "2", *SMALL)


def test_bench_line_contract(one_rank):
    """The ONE JSON line of bench.py: the keys the driver reads, the roofline object, the convergence gate."""
    d = one_rank
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["metric"] == "fwd+bwd modal-analysis passes/sec, 100k-tet ord-2 mesh, 64 modes" and d["unit"] == "passes/s"
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 0 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"] and "convergence_gate" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 4 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "stream_triad", "frac_of_stream", "in_situ", "lobpcg_spmm"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert 3000 < r["stream_triad"] < 8000 and r["in_situ"]["launches_timed"] > 0
    assert "traffic_note" in r and (r["traffic"] is None or r["traffic"] > 0)
    # round 6: the in-pass figure is `frac`, the kernel's steady state alone `frac_alone`; one hypothesis at a time, the kernel time of
    # a pass and the reference's training loop through the drop-in API ride in the line
    assert abs(r["frac_alone"] - r["achieved_alone"] / r["peak"]) < 1e-12 and r["avg_launch_ms"] > 0 and r["avg_launch_ms_alone"] > 0
    oh = d["one_hypothesis"]
    assert d["one_hypothesis_passes_per_s"] == oh["passes_per_s"] > 0 and len(oh["ms_per_pass_each"]) == oh["passes_timed"] == 12
    assert abs(oh["ms_per_pass"] * oh["passes_per_s"] - 1e3) < 1e-6 * 1e3
    assert d["kernel_stats"] is None or "error" in d["kernel_stats"] or (d["kernel_ms_per_pass"] > 0 and d["launches_per_pass"] > 50)
    api = d["api_path"]
    for key in ("cycle_1", "cycle_1_cold_start", "cycle_15"):
        assert api[key]["ms_per_epoch"] > 0 and api[key]["eigen_decompositions"] >= 1 and np.isfinite(api[key]["loss_first_last"]).all()
    assert api["cycle_15"]["ms_per_epoch"] < api["cycle_1_cold_start"]["ms_per_epoch"] and api["cycle_15"]["eigen_decompositions"] == 2
    assert "moves" in d["materials"] and d["ranks"][0]["cpu_affinity"]["bound"] in (True, False)
    # (what else runs on the box's host: load averages at start and now, one dsyevd of the Ritz step's size on one thread)
    hl = d["host_load"]
    assert len(hl["loadavg_at_start"]) == 3 and len(hl["loadavg_now"]) == 3 and 0 < hl["dsyevd_240_ms_median_of_7"] < 100
    cg = hl["cgroup_cpu"]  # (the CPU cgroup's quota and throttling counters: whole process, timed region, per leg)
    assert cg["torch_threads"] >= 1 and isinstance(cg["legs"], list) and isinstance(cg["during_the_timed_region"], dict)
    # the amortised variant (eigendecomposition every 15 passes) beside the headline, and the per-rank view of the step
    am = d["amortised"]
    assert am["eigen_decompose_cycle"] == 15 and am["unit"] == "passes/s" and am["value"] > d["value"]
    assert len(d["ranks"]) == 1 and d["ranks"][0]["hypotheses_per_step"] == 4 and d["ranks"][0]["fine_iterations"] >= 4


def test_c5_workload_line(ndev):
    """`bench.py --workload c5` (configs[4]: SpMM stress + fp64-refined solve) on a small stand-in mesh: one JSON line,
    every product form with its algorithmic bytes and bandwidth fractions, both solves converged."""
    d = _bench("--workload", "c5", "--cells", "10", "--modes", "16", "--block", "24")
    assert d["unit"] == "s" and d["higher_is_better"] is False and d["dtype"] == "f64" and d["n_gpus"] == 1
    assert "6000 tets" in d["config"]["workload"]
    s32, s64 = d["solve"]["fp32_iteration_plus_fp64_polish"], d["solve"]["with_fp64_refinement"]
    assert s32["worst_backward_error"] < 2e-6 and s64["worst_backward_error"] < 1e-10 and 1 <= s64["fp64_steps"] <= 40
    assert abs(d["value"] - s64["seconds"]) < 1e-12
    names = [p["product"] for p in d["spmm"]]
    assert any("fp32, 84" in n for n in names) and any("bf16" in n for n in names) and any("fp64" in n for n in names)
    for p in d["spmm"]:
        assert p["algorithmic_bytes"] > 0 and p["ms"] > 0 and 0 < p["frac_of_peak"] < 1
        assert abs(p["frac_of_stream"] - p["achieved_gbs"] / d["stream_triad_gbs"]) < 1e-9


def test_loss_is_bit_identical_across_runs(ndev, one_rank):
    """Concurrent hypothesis lanes on separate streams must not change a single bit of any pass: the loss sum of a
    second, identical run equals the first exactly (the check that exposed the MFMA / packed-FMA interaction of
    DESIGN.md section 4; found in round 2), here with 4 lanes on a mesh large enough for the lanes' kernels to overlap."""
    args = ["--gpus", "1", "--hyp-per-gpu", "8", "--lanes", "4", "--steps", "2", "--warmup", "0", "--cells", "12", "--modes", "32",
            "--block", "40", "--no-cpu-baseline", "--amortised-cycle", "0"]
    a, b = _bench(*args), _bench(*args)
    assert a["loss_sum_last_step"] == b["loss_sum_last_step"]
    assert [r["fine_iterations"] for r in a["ranks"]] == [r["fine_iterations"] for r in b["ranks"]]


def test_steps_without_a_join_give_the_joined_schedule_s_results(ndev, one_rank):
    """bench.py's default schedule (ModalPipeline.run_steps: a lane runs its hypotheses' consecutive steps back to back) against
    --step-barrier (all lanes join after every step): the same passes, so the same loss sum to the last bit; three steps, four
    hypotheses on two lanes (two hypotheses per lane and step)."""
    args = ["--gpus", "1", "--hyp-per-gpu", "4", "--lanes", "2", *SMALL]
    args[args.index("--steps") + 1] = "3"
    free = _bench(*args)
    joined = _bench(*args, "--step-barrier")
    assert free["loss_sum_last_step"] == joined["loss_sum_last_step"]
    # (round 6: the material moves from step to step, so the third step's loss is not the first's; with --same-material - the
    # repetition of rounds 1-5 - three steps end where one step ends)
    assert free["loss_sum_last_step"] != one_rank["loss_sum_last_step"] and "moves" in free["materials"]
    same3 = _bench(*args, "--same-material")
    same1 = _bench("--gpus", "1", "--hyp-per-gpu", "4", "--lanes", "2", *SMALL, "--same-material")
    assert same3["loss_sum_last_step"] == same1["loss_sum_last_step"]
    assert "no join between steps" in free["config"]["step_schedule"] and "join after every step" in joined["config"]["step_schedule"]
    assert free["steps"] == joined["steps"] == 3


def test_two_ranks_sharing_one_device_gloo(ndev, one_rank):
    two = _bench("--gpus", "2", "--hyp-per-gpu", "2", "--lanes", "2", "--dist-backend", "gloo", "--share-devices", *SMALL)
    assert two["n_gpus"] == 2 and "gloo" in two["collective"]
    assert abs(two["loss_sum_last_step"] / one_rank["loss_sum_last_step"] - 1) < 1e-6


def test_two_ranks_rccl(ndev, one_rank):
    if ndev < 2:
        pytest.skip("RCCL needs one device per rank; this box has one")
    two = _bench("--gpus", "2", "--hyp-per-gpu", "2", "--lanes", "2", *SMALL)
    assert two["n_gpus"] == 2 and "nccl" in two["collective"]
    assert abs(two["loss_sum_last_step"] / one_rank["loss_sum_last_step"] - 1) < 1e-6


def test_rccl_collectives_of_the_n_gt_1_path_run_with_one_rank(ndev):
    """RCCL itself on this box: the collectives bench.py's N > 1 branch makes - barrier on the rank's own device, the
    all-reduce of the scalar loss, the all-gather of the rank statistics - through the same functions, backend "nccl", in a
    process group of ONE rank (all a one-device box allows: two ranks need two devices).  Child process, so that the test
    runner itself never owns a process group."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "from diffsound_amd.pipeline import all_reduce_loss, gather_rank_stats\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "dev = torch.device('cuda', 0)\n"
        "dist.barrier(device_ids=[0])\n"
        "s = all_reduce_loss(1.25, dev)\n"
        "g = gather_rank_stats([1.0, 2.0, 3.5], dev)\n"
        "dist.barrier(device_ids=[0])\n"
        "torch.cuda.synchronize()\n"
        "print('RCCL', dist.get_backend(), s, g)\n"
        "dist.destroy_process_group()\n")
    import socket

    with socket.socket() as sk:  # a free port of this box, not a fixed one (a lingering socket of an aborted run must not
        sk.bind(("127.0.0.1", 0))  # read as an RCCL failure)
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "RCCL nccl 1.25 [[1.0, 2.0, 3.5]]" in out.stdout, out.stdout[-2000:]


def test_more_gpus_than_devices_is_an_error(ndev):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ndev + 1), *SMALL],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and "visible" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_configs3_rehearsal_64_hypotheses_on_the_benchmark_mesh(ndev):
    """configs[3] at its real shape on the one device there is: the 64 hypotheses of rng(2024) on the C3 mesh
    (105 456 tets, ord-2, 64 modes), sharded round-robin over RANKS that share the device, scalar loss all-reduced
    over gloo (RCCL refuses two ranks on one device; its path differs only in the backend string).  4 ranks x 16
    hypotheses, not 8 x 8: a GPU box of this pool kills a job with more than 6 processes on the card, and this pytest
    process is one of them - the 8-rank LAYOUT is covered on the CPU (tests/test_dist_gloo.py).  Checks: the loss sum
    equals one rank running all 64 to 1e-6, no rank idle, per-rank iteration counts and HBM footprint reported."""
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2 ** 30:
        pytest.skip("needs 80 GB of free HBM (4 ranks x lanes on the C3 mesh)")
    c3 = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    one = _bench("--gpus", "1", "--hyp-per-gpu", "64", "--lanes", "4", *c3)
    four = _bench("--gpus", "4", "--hyp-per-gpu", "16", "--lanes", "2", "--dist-backend", "gloo", "--share-devices", *c3)
    assert "105456 tets" in four["config"]["workload"] and four["n_gpus"] == 4
    assert abs(four["loss_sum_last_step"] / one["loss_sum_last_step"] - 1) < 1e-6
    ranks = four["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1, 2, 3] and all(r["hypotheses_per_step"] == 16 for r in ranks)
    assert all(r["fine_iterations"] >= 16 and r["busy_seconds"] > 0 and r["idle_fraction"] < 0.5 for r in ranks)
    assert all(0 < r["hbm_peak_allocated_gib"] < 40 for r in ranks)
    # the same 64 solves, sharded (a solve's Chebyshev interval starts from the previous hypothesis ON ITS LANE, so
    # the iteration counts are not bit-tied to the sharding: 403 against 405 in the first run)
    assert abs(sum(r["fine_iterations"] for r in ranks) / one["ranks"][0]["fine_iterations"] - 1) < 0.05
    assert abs(four["value"] - 64 / (four["ms_per_step"] * 1e-3)) < 1e-6 * four["value"]


def test_geom_workload_line(ndev):
    """`bench.py --workload geom` (the shape loop: a fresh DiffSoundObj on new vertices and a new topology per iteration, backward
    to the vertices) on a small stand-in shell: one JSON line with iterations/s, the split of an iteration and flat HBM."""
    d = _bench("--workload", "geom", "--cells", "8", "--modes", "8", "--geom-iters", "24", "--no-cpu-baseline")
    assert d["unit"] == "iterations/s" and d["value"] > 0 and d["n_gpus"] == 1 and d["iterations"] == 24
    assert abs(d["value"] * d["ms_per_iteration"] - 1e3) < 1e-6 * 1e3 and "ord-1" in d["config"]["workload"]
    assert set(d["split_ms_with_a_sync_per_stage"]) == {"vertices", "symbolic + tables", "assembly + eigensolve",
                                                        "get_vals + loss + backward + Adam"}
    assert d["hbm"]["growth_mib_torch"] <= 1.0 and d["hbm"]["growth_mib_device"] <= 64.0
    assert np.isfinite(d["loss_first_last"]).all() and d["thickness_after"] != 1.0
