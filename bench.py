#!/usr/bin/env python3
"""bench.py -- MCMC moves/sec of the EMAT local-move hot path on MI355X (BASELINE.json metric).

A "step" is one `emat_run_local_moves` pass (reference Run::run_local_moves, core/run.cpp:682-693) of
`--moves-per-part` Subrun::mcmc_sub_iteration calls on every partition part of the workload, with all part
slabs already resident in HBM.  Default workload = config C4 of SURVEY section 8(d): a seeded synthetic
100k-tip SARS-CoV-2-like EMAT (29 903 sites, HKY + skygrid) cut by the reference's tree-partitioning rule.

N > 1 (launched by torchrun, one rank per GPU): 8192 parts are requested per GPU (the partitioner's minimum part
size caps what the tree yields) and sharded across ranks in contiguous blocks;
the only cross-rank exchange is the per-cycle coalescent-grid all-reduce (<= a few KB) done
before the timed region and a 2-double all-reduce of the log-posterior totals after it.  The tree is fixed
as N grows => "scaling": "strong".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C4", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--tips", type=int, default=None, help="override the number of tips (debug)")
    ap.add_argument("--parts", type=int, default=None, help="number of partition parts requested from the partitioner (default 8192 per GPU)")
    ap.add_argument("--moves-per-part", type=int, default=1000)
    ap.add_argument("--max-part-nodes", type=int, default=0, help="not in the reference: cut parts larger than this further (0 = the reference's partitioning rule)")
    ap.add_argument("--no-lds", action="store_true")
    ap.add_argument("--no-topology", action="store_true", help="diagnostic: disable subtree-slide and SPR moves")
    ap.add_argument("--only-displace", action="store_true", help="diagnostic: only inner-node displacement moves")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="wall-time budget of the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(sc, num_parts, seed, target_seconds, t_step):
    """The CPU restatement of Delphy's algorithm (oracle/, kind = "port") timed on this box's host cores on a
    bounded sample of the same workload: the same parts, the same number of moves on every part, one thread per
    core with the parts dealt round-robin to the threads (the reference's policy is one part per thread,
    tools/delphy.cpp:130-132).  A short pilot sizes the sample to about `target_seconds` of wall time."""
    from helpers import configure, split_parts
    from oracle_ffi import OracleEngine
    cores = max(1, min(os.cpu_count() or 1, 64))
    parts, incl, seeds, root_part, ref = split_parts(sc, num_parts, seed)
    orc = OracleEngine(sc.num_sites)
    configure(orc, sc, ref, parts, incl, seeds, root_part, t_step)
    orc.recalc_derived()
    pilot = 2000
    t0 = time.perf_counter()
    orc.run_moves_per_part(pilot, threads=cores)
    rate = len(parts) * pilot / max(1e-6, time.perf_counter() - t0)
    # the pilot runs hot in cache and over-estimates the sustained rate by about 2x: size the sample for 0.45 x target
    sample_moves = int(min(400000, max(1000, 0.45 * target_seconds * rate / len(parts))))
    t0 = time.perf_counter()
    orc.run_moves_per_part(sample_moves, threads=cores)
    dt = time.perf_counter() - t0
    orc.close()
    return {"value": len(parts) * sample_moves / dt, "unit": "moves/s", "cores": cores, "kind": "port",
            "sample": "same %d parts, %d moves per part = %.3g moves (%.1f s wall on %d host threads, after a %d-move pilot); "
                      "CPU restatement of Delphy's algorithm (oracle/), not Delphy itself"
                      % (len(parts), sample_moves, len(parts) * sample_moves, dt, cores, pilot)}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # EMAT_BENCH_SHARED_GPU=1 is a plumbing check for boxes with ONE GPU: every rank uses cuda:0 and the collectives go
    # over gloo (RCCL refuses two ranks on one device).  It exercises the multi-rank control flow, not RCCL, and its
    # numbers mean nothing; the driver never sets it.
    shared_gpu = world > 1 and os.environ.get("EMAT_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the EMAT engine has no CPU fallback")

    import delphy_amd as d
    from delphy_amd.scenarios import make_scenario
    from delphy_amd.sharding import ShardedEngine

    sc = make_scenario(args.workload, num_tips=args.tips)
    # The reference cuts the tree into as many parts as it has workers (tools/delphy.cpp:130-132); here a worker is a
    # wavefront slot, ~4 000 per GPU, so the request grows with the number of GPUs.  The tree is the same at every N
    # ("strong" scaling); the partitioner's minimum part size (10 branches) caps what a 100k-tip tree can yield.
    if args.parts is None:
        args.parts = min(8192 * world, 16384)   # this tree yields ~13 000 parts at most; asking for more only unbalances them
    allreduce = None
    if shared_gpu:
        def allreduce(arr, op):
            t = torch.from_numpy(np.ascontiguousarray(arr))
            dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op])
            return t.numpy()
    eng = ShardedEngine(sc, num_parts=args.parts, seed=20261001, rank=rank, world=world, device=local_rank, use_lds=not args.no_lds, allreduce=allreduce, max_part_nodes=args.max_part_nodes)
    eng.topology = not args.no_topology
    eng.only_displace = args.only_displace
    eng.setup()   # partition, upload this rank's parts, exchange the coalescent grid, recalc derived quantities

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.backend.synchronize()

    for _ in range(args.warmup):
        eng.backend.run_moves_per_part(args.moves_per_part)
    barrier()
    stats0 = eng.local_stats()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.backend.run_moves_per_part(args.moves_per_part)
        kernel_ms.append(None)
    barrier()
    dt = time.perf_counter() - t0
    # per-launch kernel time from HIP events on the engine's own stream (only the last launch's events survive
    # back-to-back launches, so re-time K launches one by one OUTSIDE the timed region for the roofline figure)
    ev_ms = []
    for _ in range(min(args.steps, 3)):
        eng.backend.run_moves_per_part(args.moves_per_part)
        eng.backend.synchronize()
        ev_ms.append(eng.backend.last_run_ms())
    stats1 = eng.local_stats()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_parts = eng.total_parts
    local_parts = eng.num_local_parts
    moves_timed_all = total_parts * args.moves_per_part * args.steps
    value = moves_timed_all / dt
    log_G, log_prior = eng.global_totals()

    # roofline of the dominant kernel (k_run_moves): algorithmic bytes counted by the kernel per executed move
    launches = args.steps + len(ev_ms)
    bytes_per_launch = (stats1["algorithmic_bytes"] - stats0["algorithmic_bytes"]) / max(1, launches)
    avg_ms = float(np.mean(ev_ms)) if ev_ms else float("nan")
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    # the committed PMC figure was collected on the default single-GPU workload: only quote it for that one
    if os.path.exists(pmc_path) and world == 1 and args.workload == "C4" and args.tips is None and args.parts == 8192 and args.moves_per_part == 1000 and args.max_part_nodes == 0 and not (args.no_topology or args.only_displace or args.no_lds):
        try:
            traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    bad = stats1["bad_parts"]

    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(sc, args.parts, 20261001, args.cpu_seconds, eng.t_step)

    if rank == 0:
        out = {
            "metric": "MCMC moves/sec on 100k-tip SARS-CoV-2 EMAT at 1/2/4/8 MI355X",
            "value": value,
            "unit": "moves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: synthetic %d-tip EMAT, %d sites, HKY(kappa=5) + %s, %d partition parts (%d nodes), %d moves/part/step, move mix 7.5/7.5/15/1/1"
                            % (sc.name, sc.num_tips, sc.num_sites, {0: "constant pop", 1: "exponential-growth coalescent", 2: "skygrid"}[sc.pop.kind],
                               total_parts, sc.tree.num_nodes, args.moves_per_part),
                "parts_per_gpu": local_parts, "max_part_nodes": args.max_part_nodes,
                "lds_staging": not args.no_lds,
                "parallelism": "parts sharded over %d GPU(s), one wavefront per part" % world,
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_run_moves", "kernel_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch},
            "cpu_baseline": cpu_base,
            "check": {"log_G": log_G, "log_augmented_coalescent_prior": log_prior, "parts_stopped": bad},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
