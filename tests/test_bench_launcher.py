"""bench.py --gpus N starts N ranks itself when no launcher did (CPU: the ranks stop at the missing GPU, loudly)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_flag_spawns_that_many_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side check of the launcher (on a GPU box the -m gpu test runs the real thing)")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"EMAT_BENCH_SHARED_GPU": "1"})
    assert r.returncode != 0                              # no GPU here: every rank must refuse, not fall back
    # (the launcher stops the other rank as soon as one has refused: that one may or may not have got its message out, but the
    # launcher's report names both)
    assert ("rank 0 of 2" in r.stderr or "rank 1 of 2" in r.stderr) and "local_rank: 0" in r.stderr and "local_rank: 1" in r.stderr, r.stderr[-2000:]
    assert '"metric"' not in r.stdout


def test_gpus_flag_refuses_more_ranks_than_gpus():
    import torch
    r = _run(["--gpus", "64"], {})
    assert r.returncode != 0 and "this node has %d GPU" % torch.cuda.device_count() in r.stderr     # (counted in sysfs, or by a child: the launcher itself never touches a GPU)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_on_the_one_gpu_of_the_test_box():
    """`python bench.py --gpus 2` as the driver would type it, on a one-GPU box (both ranks share cuda:0, collectives
    over gloo: the control flow of a 2-rank run, not RCCL): one JSON line, n_gpus 2, both ranks' parts and times."""
    import json
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--inclusive-cycles", "4", "--secondary", ""], {"EMAT_BENCH_SHARED_GPU": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    # whole sharded cycles are timed at N > 1 too, by phase, on both decompositions
    inc = out["inclusive"]
    assert inc["cycles"] == 4 and inc["value"] > 0 and inc["exchange_bytes_per_cycle_all_ranks"] > 1e6 and sum(inc["parts_per_rank_last_cycle"]) == inc["parts_of_the_last_cycle"]
    phases = inc["ms_per_cycle_by_phase_rank_0"]
    assert any(k.startswith("1 repartition") for k in phases) and any(k.startswith("2b the pass") for k in phases) and any("all-gather" in k for k in phases) and any(k.startswith("7 apply") for k in phases)
    assert abs(sum(phases.values()) - inc["ms_per_cycle"]) < 0.25 * inc["ms_per_cycle"]          # the phases are the cycle
    assert inc["fixed_partition"]["parts_requested"] == 8192 and inc["parts_requested"] == 16384
    dec = out["decompositions"]
    assert dec["fixed_partition"]["parts"] < dec["grown_partition"]["parts"] and dec["fixed_partition"]["value"] > 0 and dec["grown_partition"]["value"] == out["value"]
    assert out["n_gpus"] == 2 and len(out["per_rank"]["ms_per_step"]) == 2
    assert sum(out["per_rank"]["parts"]) > 7000 and out["check"]["parts_stopped"] == 0
    assert out["value"] > 0
    # the run checks itself: what the two ranks computed is what one rank computes (bench.py scale_check)
    sc = out["scale_check"]
    assert sc["ok"] and sc["ranks_seen"] == 2 and sc["parts_reported"] == sc["parts_of_the_run"] == sum(out["per_rank"]["parts"])
    assert sc["parts_with_other_move_or_draw_counts"] == 0 and sc["parts_whose_trees_differ"] == 0 and sc["parts_verified"] >= 8 and sc["max_rel_err"] < 1e-9


@pytest.mark.gpu
def test_eight_ranks_on_the_one_gpu_of_the_test_box():
    """`python bench.py --gpus 8`, the command of the driver's scaling run, with all eight ranks on the one GPU of the test box (gloo):
    it must finish within ten minutes, check itself (scale_check: what the eight ranks computed is what one rank computes), time every
    rank's setup, and carry the C5 series as `secondary`."""
    import json
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-inclusive", "--secondary-steps", "1"], {"EMAT_BENCH_SHARED_GPU": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and len(out["per_rank"]["ms_per_step"]) == 8 and len(out["per_rank"]["setup_s"]) == 8 and max(out["per_rank"]["setup_s"]) < 120
    sc = out["scale_check"]
    assert sc["ok"] and sc["ranks_seen"] == 8 and sc["parts_reported"] == sc["parts_of_the_run"] == sum(out["per_rank"]["parts"])
    sec = out["secondary"]
    assert sec["workload"].startswith("C5") and len(sec["per_rank"]["parts"]) == 8 and sum(sec["per_rank"]["parts"]) > 70000 and sec["value"] > 0 and sec["parts_stopped_on_rank_0"] == 0


@pytest.mark.gpu
def test_two_ranks_on_two_gpus_over_rccl():
    """`python bench.py --gpus 2` WITHOUT EMAT_BENCH_SHARED_GPU: one rank per GPU, collectives over RCCL -- what the driver's scaling run
    types.  Skipped on a one-GPU box (every lease so far); on the first node with two GPUs it runs by itself and must pass its own
    scale_check with rccl_world == 2."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the test box has one)")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--secondary-steps", "1"], {})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and len(out["per_rank"]["ms_per_step"]) == 2
    sc = out["scale_check"]
    assert sc["ok"] and sc["ranks_seen"] == 2 and sc["rccl_world"] == 2 and sc["parts_with_other_move_or_draw_counts"] == 0 and sc["parts_whose_trees_differ"] == 0
