#!/usr/bin/env python3
"""bench.py -- MCMC moves/sec of the EMAT local-move hot path on MI355X (BASELINE.json metric).

A "step" is one `emat_run_local_moves` pass (reference Run::run_local_moves, core/run.cpp:682-693) of
`--moves-per-part` Subrun::mcmc_sub_iteration calls on every partition part of the workload, with all part
slabs already resident in HBM.  Default workload = config C4 of SURVEY section 8(d): a seeded synthetic
100k-tip SARS-CoV-2-like EMAT (29 903 sites, HKY + skygrid) cut by the reference's tree-partitioning rule.

N > 1: one rank per GPU over RCCL.  Launched either by the driver (`python -m torch.distributed.run ... bench.py --gpus N`:
RANK / WORLD_SIZE come from the environment) or as plain `python bench.py --gpus N`: the parent then starts the N ranks
itself through torch.distributed.run BEFORE touching any GPU, relays rank 0's JSON line and exits with the children's code.
The SAME partition at every N (8192 parts requested, 7 955 obtained),
sharded across ranks in contiguous blocks; the only cross-rank exchange is the per-cycle coalescent-grid all-reduce
(<= a few KB) done before the timed region and a 2-double all-reduce of the log-posterior totals after it.
Tree and partition are fixed as N grows => "scaling": "strong".
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


def device_code_sha16():
    """Identifies the device code a PMC profile belongs to: the id csrc/Makefile stamps into the library (emat_build_id),
    recomputed from the sources the kernels are compiled from."""
    from delphy_amd.engine import source_build_id
    return source_build_id()


DEFAULT_PARTS = {"C1": 8, "C2": 128, "C3": 1024, "C4": 8192, "C5": 81920}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C4", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--tips", type=int, default=None, help="override the number of tips (debug)")
    ap.add_argument("--parts", type=int, default=None, help="number of partition parts requested from the partitioner; default: 8192 at C4, 81920 at C5 (about 25 nodes per part), "
                    "and at least 8192 per GPU -- the reference cuts as many parts as it has workers (tools/delphy.cpp:130-132), here two per wavefront slot; "
                    "the partitioner's floor of 10 nodes per part caps what a tree yields (13 140 parts at C4)")
    ap.add_argument("--moves-per-part", type=int, default=None, help="moves per part and step; default 1000 on one GPU, and on N > 1 what keeps the moves of a step the same as on one GPU")
    ap.add_argument("--max-part-nodes", type=int, default=-1, help="not in the reference: parts larger than this get further, randomly drawn cut nodes at every repartition "
                                                                    "(-1 = the run driver's default, three times the mean part size; 0 = the reference's partitioning rule exactly)")
    ap.add_argument("--secondary", default="C5", help="a second workload measured the same way and reported as `secondary` (default C5, the one with enough parts for eight GPUs; '' = none)")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--inclusive-cycles", type=int, default=210, help="whole cycles of the `inclusive` figure (the reference redraws its partition stencils every 200 cycles)")
    ap.add_argument("--no-lds", action="store_true")
    ap.add_argument("--no-topology", action="store_true", help="diagnostic: disable subtree-slide and SPR moves")
    ap.add_argument("--only-displace", action="store_true", help="diagnostic: only inner-node displacement moves")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time budget of each CPU baseline sample")
    ap.add_argument("--verify-parts", type=int, default=4, help="N > 1: parts per rank whose final trees are compared with a one-GPU run of the same partition on rank 0 "
                    "(outside the timed region; the counters of ALL parts are compared whatever this says; 0 = no check)")
    ap.add_argument("--no-decompositions", action="store_true", help="skip the second decomposition (N = 1: the partition N >= 2 run; N >= 2: N = 1's partition)")
    ap.add_argument("--no-inclusive", action="store_true", help="skip the host-cycle-inclusive figure (repartition + moves + reassemble through the run driver)")
    args = ap.parse_args()
    args.parts_auto = args.parts is None
    if args.parts is None:
        args.parts = DEFAULT_PARTS[args.workload]
    args.moves_auto = args.moves_per_part is None
    if args.moves_per_part is None:
        args.moves_per_part = 1000
    return args


def count_gpus_without_touching_them():
    """Number of GPUs of this node read from the KFD topology in sysfs (a node with a non-zero `simd_count` is a GPU), cut down
    by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when those are set: no HIP or HSA call, so the process that asks stays free
    of any GPU state (torch.cuda.device_count() falls through to hipGetDeviceCount on ROCm, which initialises the runtime).
    -1 when the topology cannot be read (no amdgpu driver: then the question is asked of a short-lived child instead)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            props = dict(ln.split() for ln in open(os.path.join(base, node, "properties")) if len(ln.split()) == 2)
            n += 1 if int(props.get("simd_count", "0")) > 0 else 0
    except OSError:
        return -1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks (one per GPU) as children through
    torch.distributed.run and relay what they print.  Nothing in this process initialises the GPU -- the devices are counted
    in sysfs, or by a short-lived child when sysfs has no KFD topology -- and the children are started as ordinary
    subprocesses, never by exec."""
    import socket
    import subprocess
    have = count_gpus_without_touching_them()
    if have < 0:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
        have = int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
    if have < args.gpus and os.environ.get("EMAT_BENCH_SHARED_GPU") != "1":
        raise SystemExit("bench.py --gpus %d: this node has %d GPU(s) (EMAT_BENCH_SHARED_GPU=1 runs the ranks on one GPU over gloo as a plumbing check)" % (args.gpus, have))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stdout.write(ln)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
    sys.stdout.flush()
    return rc


def _time_oracle(sc, num_parts, seed, target_seconds, t_step, threads, pilot):
    """Oracle (CPU restatement) on `num_parts` parts of `sc`: the same number of moves on every part, parts dealt round-robin
    to `threads` host threads; a short pilot sizes the sample to about `target_seconds` of wall time."""
    from helpers import configure, split_parts
    from oracle_ffi import OracleEngine
    parts, incl, seeds, root_part, ref = split_parts(sc, num_parts, seed)
    orc = OracleEngine(sc.num_sites)
    configure(orc, sc, ref, parts, incl, seeds, root_part, t_step)
    orc.recalc_derived()
    t0 = time.perf_counter()
    orc.run_moves_per_part(pilot, threads=threads)
    rate = len(parts) * pilot / max(1e-6, time.perf_counter() - t0)
    # the pilot runs hot in cache and over-estimates the sustained rate by about 2x: size the sample for 0.45 x target
    sample_moves = int(min(50_000_000, max(pilot, 0.45 * target_seconds * rate / len(parts))))
    t0 = time.perf_counter()
    orc.run_moves_per_part(sample_moves, threads=threads)
    dt = time.perf_counter() - t0
    orc.close()
    return len(parts), sample_moves, dt


def cpu_baseline(sc, num_parts, seed, target_seconds, t_step):
    """The CPU restatement of Delphy's algorithm (oracle/, kind = "port") timed on this box's host cores on bounded
    samples of the same workload.  `value` = the GPU's own partition replayed on the CPU (the same ~7 955 small parts,
    dealt round-robin to one thread per core).  `policies` adds the two figures SURVEY 8(d) asks for: the reference's own
    policy of as many parts as cores (tools/delphy.cpp:130-132) on the same tree, and config C1 on a single thread."""
    from delphy_amd.scenarios import make_scenario
    cores = max(1, min(os.cpu_count() or 1, 64))
    # three samples, the median quoted and the spread beside it: one 5-second sample varied by 25 % from box to box and run to run
    samples = sorted((_time_oracle(sc, num_parts, seed, target_seconds / 2.0, t_step, cores, 2000) for _ in range(3)), key=lambda x: x[0] * x[1] / x[2])
    n, moves, dt = samples[1]
    out = {"value": n * moves / dt, "unit": "moves/s", "cores": cores, "kind": "port", "value_min": samples[0][0] * samples[0][1] / samples[0][2], "value_max": samples[2][0] * samples[2][1] / samples[2][2],
           "sample": "GPU partition replayed on the CPU, three times (median quoted, min and max beside it): same %d parts, %d moves per part = %.3g moves (%.1f s wall on %d host threads, after a 2000-move pilot); "
                     "CPU restatement of Delphy's algorithm (oracle/), not Delphy itself" % (n, moves, n * moves, dt, cores)}
    n2, moves2, dt2 = _time_oracle(sc, cores, seed, target_seconds, t_step, cores, 20000)
    c1 = make_scenario("C1")
    n3, moves3, dt3 = _time_oracle(c1, 1, seed, min(target_seconds, 5.0), c1.default_t_step(), 1, 20000)
    out["policies"] = {
        "parts_equal_cores": {"value": n2 * moves2 / dt2, "unit": "moves/s", "cores": cores,
                              "sample": "reference policy num_parts = cores: same tree in %d parts, %d moves per part (%.1f s wall)" % (n2, moves2, dt2)},
        "c1_single_thread": {"value": n3 * moves3 / dt3, "unit": "moves/s", "cores": 1,
                             "sample": "config C1 (100 tips, 30 000 sites, one part), %d moves on one thread (%.1f s wall)" % (moves3, dt3)},
    }
    return out


def inclusive_cycles(sc, args, cycles=210):
    """Whole cycles through the host run driver (emat_run_do_mcmc_steps), the reference's default 50 x nodes local moves per
    cycle (run.cpp:669-672), wall clock, cycle by cycle over more than one stencil period of the reference (its partition stencils
    are redrawn every 200 cycles, run.cpp:87-108, and the parts they delimit drift apart in size while they are in use).  Three ways:
    with the whole tree resident in HBM (the host draws the partition on the topology, kernels cut the part slabs
    and gather them back) under the run driver's default part-size limit; the same under the reference's partitioning rule exactly
    (emat_run_set_max_part_nodes(0), fewer cycles: a cycle then lasts as long as the chain of its largest part); and with the tree on
    the host (subtree build -> slab encode -> H2D -> moves -> D2H -> decode -> reassemble).  Reported beside `value`, never as it."""
    import numpy as np
    import delphy_amd as d
    per_cycle = 50 * sc.tree.num_nodes

    def one(device_tree, limit, n, parts=None, download=False):
        b = d.EmatBackend(sc.num_sites)
        run = d.EmatRun(b, sc.tree, sc.ref, 20261001)
        run.set_num_parts(args.parts if parts is None else parts)
        run.set_max_part_nodes(limit)
        run.set_hky(sc.mu, sc.kappa, sc.pi)
        run.set_pop_model(sc.pop)
        if device_tree:
            run.set_device_tree(True)
        run.do_mcmc_steps(per_cycle, per_cycle)          # warm-up cycle: allocations, first launch, tree upload, the first ten stencils
        rows = []
        t0 = time.perf_counter()
        for _ in range(n):                               # (one call per cycle only to time the cycles apart; the driver repartitions at every cycle boundary either way)
            t1 = time.perf_counter(); run.do_mcmc_steps(per_cycle, per_cycle)
            if download:
                b.tree_download()                          # the whole tree and the reference sequence into host arrays (emat_tree_download), every cycle
            ms = (time.perf_counter() - t1) * 1e3
            st = run.partition_stats()
            rows.append((ms, st["num_parts"], st["largest_part_nodes"], st["extra_cuts"]))
        dt = time.perf_counter() - t0
        lim = run.partition_stats()["max_part_nodes"]
        run.close(); b.close()
        a = np.array(rows); k = max(1, n // 10)
        p10, p50, p90 = (float(np.percentile(a[:, 0], q)) for q in (10, 50, 90))
        return {"value": n * per_cycle / dt, "unit": "moves/s", "cycles": n, "ms_per_cycle": dt / n * 1e3, "ms_per_cycle_min": float(a[:, 0].min()), "ms_per_cycle_max": float(a[:, 0].max()),
                "ms_per_cycle_p10": p10, "ms_per_cycle_p50": p50, "ms_per_cycle_p90": p90, "p90_over_p10": p90 / p10, "max_part_nodes": lim,
                "by_tenth_of_the_run": [{"ms_per_cycle": float(a[i: i + k, 0].mean()), "parts": float(a[i: i + k, 1].mean()), "largest_part_nodes": int(a[i: i + k, 2].max()),
                                         "cut_nodes_added_by_the_limit": float(a[i: i + k, 3].mean())} for i in range(0, n, k)]}

    dev = one(True, args.max_part_nodes, cycles)
    ref_rule = one(True, 0, max(10, cycles // 5)) if args.max_part_nodes != 0 else None
    host = one(False, args.max_part_nodes, max(5, cycles // 20))
    dev_dl = one(True, args.max_part_nodes, max(10, cycles // 5), download=True)
    out = dict(dev, moves_per_cycle=per_cycle,
               what="emat_run_do_mcmc_steps with the tree resident in HBM: stencil + partition on the host's copy of the topology, part slabs cut and "
                    "gathered back by kernels, %d local moves per cycle (no global moves), wall clock, %d successive cycles (the reference redraws its stencils every 200)" % (per_cycle, cycles),
               host_tree=dict(host, what="the same cycles with the tree on the host: subtree build + slab encode + H2D + moves + D2H + decode + reassemble"),
               resident_tree_downloaded_every_cycle=dict(dev_dl, what="the resident-tree cycles with emat_tree_download after every cycle: what a Run that wants its Phylo_tree on the host after every "
                                                                       "cycle (logging, global moves that read the tree) pays on top -- the tree stays authoritative on the device, the host gets a copy"))
    if args.parts_auto and args.workload == "C4":
        # what N >= 2 cut (bench.py requests 8192 parts per GPU; the partitioner's floor of ten nodes per part caps C4 at about 13 000): whole cycles of ONE GPU on that
        # partition, so that a scaling curve of `inclusive` has its x 1 on the same decomposition as its other points
        out["grown_partition"] = dict(one(True, args.max_part_nodes, max(10, cycles // 5), parts=8 * args.parts),
                                      what="the same cycles with %d parts requested, the partition bench.py --gpus N >= 2 runs" % (8 * args.parts))
    if ref_rule is not None:
        out["reference_partition_rule"] = dict(ref_rule, what="the same cycles with emat_run_set_max_part_nodes(0): the reference's partitioning rule exactly, whose parts drift apart in size between "
                                                                "two redraws of the stencils; every part makes the same number of moves, so a cycle lasts as long as the chain of its largest part")
    return out


def inclusive_sharded(sc, args, parts, cycles, world, rank, local_rank, shared_gpu, dist, torch):
    """Whole SHARDED cycles (N > 1), the counterpart of `inclusive_cycles`: every rank keeps the whole tree in its HBM and a contiguous
    block of the parts; one cycle = repartition (every rank draws the same partition, cuts its own block) -> the reference's 50 x nodes
    local moves -> gather of the local parts, all-gather of what they own (the tree's lists, once per cycle), apply, totals
    (delphy_amd.sharding.ShardedEngine.cycle = reference Run::do_mcmc_steps without global moves, run.cpp:622-657).  Wall clock per
    cycle, MAX over ranks; rank 0's time by phase.  Over RCCL the node exchange stays in device buffers; with EMAT_BENCH_SHARED_GPU
    (all ranks on one GPU, gloo) it goes through host buffers and the times mean nothing -- only that the path runs."""
    import numpy as np
    from delphy_amd.sharding import ShardedEngine
    dev = "cpu" if shared_gpu else "cuda"
    allreduce = allgather = None
    if shared_gpu:
        def allreduce(arr, op):
            t = torch.from_numpy(np.ascontiguousarray(arr))
            dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op])
            return t.numpy()

        def allgather(buf):
            sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(sizes, torch.tensor([buf.shape[0]], dtype=torch.int64))
            n = [int(x.item()) for x in sizes]
            padded = torch.zeros(max(max(n), 1), dtype=torch.uint8); padded[: buf.shape[0]] = torch.from_numpy(np.ascontiguousarray(buf))
            out = [torch.zeros_like(padded) for _ in range(world)]
            dist.all_gather(out, padded)
            return [o[: n[r]].numpy() for r, o in enumerate(out)]
    eng = ShardedEngine(sc, num_parts=parts, seed=20261001, rank=rank, world=world, device=local_rank, use_lds=not args.no_lds, allreduce=allreduce, allgather_bytes=allgather,
                        max_part_nodes=args.max_part_nodes, device_tree=True)
    eng.topology = not args.no_topology
    eng.only_displace = args.only_displace
    per_cycle = 50 * sc.tree.num_nodes
    eng.cycle(per_cycle)                      # warm-up cycle: tree upload, allocations, the first ten stencils, communicator warm-up
    dist.barrier(); torch.cuda.synchronize()
    eng.spans = {}
    rows, nbytes = [], []
    t0 = time.perf_counter()
    for _ in range(cycles):
        t1 = time.perf_counter()
        eng.cycle(per_cycle)
        rows.append((time.perf_counter() - t1) * 1e3)
        nbytes.append(getattr(eng, "last_exchange_bytes", 0))
    dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    total_parts, local_parts = eng.total_parts, eng.num_local_parts
    spans = {k: v / cycles * 1e3 for k, v in sorted(eng.spans.items())}
    mine = torch.tensor([spans.get(k, 0.0) for k in sorted(spans)] + [float(local_parts)], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    eng.close()
    a = np.array(rows)
    return {"value": cycles * per_cycle / dt, "unit": "moves/s", "cycles": cycles, "moves_per_cycle": per_cycle, "ms_per_cycle": dt / cycles * 1e3,
            "ms_per_cycle_p10": float(np.percentile(a, 10)), "ms_per_cycle_p50": float(np.percentile(a, 50)), "ms_per_cycle_p90": float(np.percentile(a, 90)),
            "parts_requested": parts, "parts_of_the_last_cycle": total_parts, "max_part_nodes": args.max_part_nodes,
            "exchange_bytes_per_cycle_all_ranks": float(np.mean(nbytes)),
            "ms_per_cycle_by_phase_rank_0": spans,
            "ms_per_cycle_by_phase_max_over_ranks": {k: max(float(e[i].item()) for e in every) for i, k in enumerate(sorted(spans))},
            "parts_per_rank_last_cycle": [int(e[-1].item()) for e in every],
            "what": "ShardedEngine.cycle with the tree resident in every rank's HBM: every rank draws the same partition and cuts its own block of parts, %d local moves per cycle over "
                    "all ranks (no global moves), gather of the local parts, one all-gather of the nodes they own (%s), apply, all-reduce of the totals; wall clock, %d successive cycles, "
                    "MAX over ranks" % (per_cycle, "host buffers over gloo, all ranks on ONE GPU: a plumbing check, the times mean nothing" if shared_gpu else "device buffers over RCCL", cycles)}


def measure_secondary(args, world, rank, local_rank, shared_gpu, allreduce, dist, torch):
    """`args.secondary` (C5) through the same timed region as the headline: W warm-up passes, K passes between barriers, the MAX over
    ranks of the time, all parts of the run / that time."""
    from delphy_amd.scenarios import make_scenario
    t0 = time.perf_counter()
    sc2 = make_scenario(args.secondary)
    parts = max(DEFAULT_PARTS.get(args.secondary, 8192), 8192 * world)
    return measure_resident(args, sc2, parts, 1000, args.secondary_steps, world, rank, local_rank, shared_gpu, allreduce, dist, torch, t0)


def measure_resident(args, sc2, parts, moves, steps, world, rank, local_rank, shared_gpu, allreduce, dist, torch, t0=None, moves_of_a_step=None):
    """A workload / decomposition through the same timed region as the headline: one warm-up pass, `steps` passes between barriers, the
    MAX over ranks of the time, all parts of the run / that time.  `moves_of_a_step`: scale the moves per part so that a step has that
    many moves whatever the partition yields (the headline's rule at N > 1)."""
    import numpy as np
    from delphy_amd.sharding import ShardedEngine
    t0 = time.perf_counter() if t0 is None else t0
    eng = ShardedEngine(sc2, num_parts=parts, seed=20261001, rank=rank, world=world, device=local_rank, use_lds=not args.no_lds, allreduce=allreduce, max_part_nodes=args.max_part_nodes)
    eng.topology = not args.no_topology
    eng.only_displace = args.only_displace
    eng.setup()
    setup_s = time.perf_counter() - t0
    if moves_of_a_step is not None:
        moves = max(1, int(round(moves_of_a_step / eng.total_parts)))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.backend.synchronize()

    eng.backend.run_moves_per_part(moves)
    barrier()
    t1 = time.perf_counter()
    for _ in range(steps):
        eng.backend.run_moves_per_part(moves)
    barrier()
    dt = time.perf_counter() - t1
    rank_ms, rank_parts, rank_setup = [dt / steps * 1e3], [eng.num_local_parts], [setup_s]
    if world > 1:
        mine = torch.tensor([rank_ms[0], float(eng.num_local_parts), setup_s], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms, rank_parts, rank_setup = [float(e[0].item()) for e in every], [int(e[1].item()) for e in every], [float(e[2].item()) for e in every]
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    bad = eng.local_stats()["bad_parts"]
    total = eng.total_parts
    eng.close()
    return {"workload": "%s: synthetic %d-tip EMAT, %d sites, %d partition parts (%d nodes), %d moves/part/step" % (sc2.name, sc2.num_tips, sc2.num_sites, total, sc2.tree.num_nodes, moves),
            "value": total * moves * steps / dt, "unit": "moves/s", "steps": steps, "warmup": 1, "ms_per_step": dt / steps * 1e3, "parts_requested": parts, "parts": total, "moves_per_part": moves,
            "scaling": "strong", "per_rank": {"ms_per_step": rank_ms, "parts": rank_parts, "setup_s": rank_setup}, "parts_stopped_on_rank_0": bad}


def _part_digest(backend, p):
    """What a part's chain left behind, as numbers that survive a float64 tensor: 48 bits of a hash of everything discrete (topology,
    sites, states, interval endpoints), the sum of its node times, its log G and augmented prior, its move and random-draw counts."""
    import hashlib
    import numpy as np
    t = backend.part_download(p)
    h = hashlib.sha256()
    for a in (t.parent, t.child0, t.child1, t.mut_offset, t.mut_site, t.mut_from, t.mut_to, t.miss_offset, t.miss_start, t.miss_end, t.mfs_offset, t.mfs_site, t.mfs_state):
        h.update(np.ascontiguousarray(a).tobytes())
    _, _, log_G, log_aug = backend.part_derived(p, t.num_nodes)
    st = backend.part_stats(p)
    return [float(int.from_bytes(h.digest()[:6], "little")), float(np.sum(t.t)), log_G, log_aug, float(st["moves_done"]), float(st["rng_draws"])]


def scale_check(args, sc, eng, passes, world, rank, shared_gpu, dist, torch):
    """N > 1: is what the N ranks computed what ONE GPU computes?  Every chain is a deterministic function of its part, its seed and
    the coalescent tables, so after the same number of passes a part must be the same part wherever it ran.  Every rank reports the
    move and random-draw counters of all its parts and the digests of `--verify-parts` of them (plus the root part); rank 0 then
    runs the whole partition alone on its own GPU, outside the timed region, and compares: counters of every part exactly,
    everything discrete of the sampled parts exactly, times / log G / prior to rounding (the cross-rank sums of the coalescent grid
    associate differently from the one-rank sum: last-place differences in the tables)."""
    import numpy as np
    from delphy_amd.sharding import ShardedEngine
    dev = "cpu" if shared_gpu else "cuda"
    nloc, lo = eng.num_local_parts, eng.part_lo
    k = max(0, min(args.verify_parts, nloc))
    sample = sorted(set([int(round(i * (nloc - 1) / max(1, k - 1))) for i in range(k)] + ([eng.local_root] if eng.local_root >= 0 else [])))
    width = max(1, args.verify_parts) + 1
    dig = np.full((width, 7), -1.0)
    for j, p in enumerate(sample):
        dig[j] = [float(lo + p)] + _part_digest(eng.backend, p)
    cap = (eng.total_parts + world - 1) // world + 1
    ctr = np.full((cap, 3), -1.0)
    for p in range(nloc):
        st = eng.backend.part_stats(p)
        ctr[p] = [float(lo + p), float(st["moves_done"]), float(st["rng_draws"])]
    mine = torch.from_numpy(np.concatenate([dig.reshape(-1), ctr.reshape(-1)])).to(dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    out = None
    if rank == 0:
        ref = ShardedEngine(sc, num_parts=args.parts, seed=20261001, rank=0, world=1, device=eng.backend_device, use_lds=not args.no_lds, max_part_nodes=args.max_part_nodes)
        ref.topology, ref.only_displace = eng.topology, eng.only_displace
        ref.setup()
        for _ in range(passes):
            ref.backend.run_moves_per_part(args.moves_per_part)
        ref.backend.synchronize()
        ref_ctr = np.array([[ref.backend.part_stats(p)["moves_done"], ref.backend.part_stats(p)["rng_draws"]] for p in range(ref.num_local_parts)], np.float64)
        seen, ranks_seen, ctr_bad, verified, discrete_bad, max_rel = np.zeros(ref.num_local_parts, bool), 0, 0, 0, 0, 0.0
        for r, e in enumerate(every):
            e = e.cpu().numpy()
            d_r, c_r = e[: width * 7].reshape(width, 7), e[width * 7:].reshape(cap, 3)
            c_r = c_r[c_r[:, 0] >= 0]
            ranks_seen += 1 if c_r.shape[0] > 0 else 0
            ids = c_r[:, 0].astype(np.int64)
            seen[ids] = True
            ctr_bad += int(np.sum((c_r[:, 1] != ref_ctr[ids, 0]) | (c_r[:, 2] != ref_ctr[ids, 1])))
            for row in d_r[d_r[:, 0] >= 0]:
                want = _part_digest(ref.backend, int(row[0]))
                verified += 1
                discrete_bad += int(row[1] != want[0] or row[5] != want[4] or row[6] != want[5])
                for a, b in zip(row[2:5], want[1:4]):
                    max_rel = max(max_rel, abs(a - b) / max(1.0, abs(a), abs(b)))
        out = {"ranks_seen": ranks_seen, "rccl_world": dist.get_world_size(), "backend": dist.get_backend(), "passes_compared": passes,
               "parts_of_the_run": int(ref.num_local_parts), "parts_reported": int(seen.sum()), "parts_with_other_move_or_draw_counts": ctr_bad,
               "parts_verified": verified, "parts_whose_trees_differ": discrete_bad, "max_rel_err": max_rel,
               "ok": bool(seen.all() and ranks_seen == world and ctr_bad == 0 and discrete_bad == 0 and max_rel < 1e-9),
               "reference": "the same partition and passes on rank 0's GPU alone (world = 1), outside the timed region: move and random-draw counts of every part, "
                            "topology / sites / states / intervals (hashed) and node times, log G, augmented prior of the sampled parts and of the root part"}
        ref.close()
    dist.barrier()
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
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
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the EMAT engine has no CPU fallback (rank %d of %d)" % (rank, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import delphy_amd as d
    from delphy_amd.scenarios import make_scenario
    from delphy_amd.sharding import ShardedEngine
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs" % (args.gpus, world), file=sys.stderr)
    # a prebuilt library that was not compiled from the sources beside it would make every number here describe other code
    build_id = d.library_build_id()
    if build_id != d.source_build_id() and os.environ.get("EMAT_ALLOW_STALE_LIB") != "1":
        raise SystemExit("bench.py: %s was built from device code %s, the sources are %s: rebuild (python -c 'import __graft_entry__ as g; g.build()')"
                         % (d.library_path(), build_id, d.source_build_id()))

    # The reference cuts the tree into as many parts as it has workers (tools/delphy.cpp:130-132).  Here a worker is a wavefront slot, 4 096
    # per GPU, and a pass wants about two parts per slot (with one, a rank's pass is as long as its slowest chain): the request grows with
    # the number of GPUs, the moves of a step stay what they are on one GPU.  Same tree, same total work at every N: "strong" scaling.
    base_parts = args.parts
    if args.parts_auto:
        args.parts = max(args.parts, 8192 * world)
    t_setup0 = time.perf_counter()
    sc = make_scenario(args.workload, num_tips=args.tips)
    # The reference cuts the tree into as many parts as it has workers (tools/delphy.cpp:130-132); here a worker is a
    # wavefront slot, ~4 000 per GPU, so the request grows with the number of GPUs.  The tree is the same at every N
    # ("strong" scaling); the partitioner's minimum part size (10 branches) caps what a 100k-tip tree can yield.
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
    setup_s = time.perf_counter() - t_setup0     # every rank generates the tree and draws the partition itself (same seed: no communication)
    if args.moves_auto and args.parts != base_parts:
        args.moves_per_part = max(1, int(round(1000.0 * base_parts / eng.total_parts)))   # about the moves a step has on one GPU, over the finer partition

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
    rank_ms = [dt / args.steps * 1e3]
    rank_kernel_ms = [float(np.mean(ev_ms)) if ev_ms else float("nan")]
    rank_parts = [eng.num_local_parts]
    rank_setup_s = [setup_s]
    if world > 1:
        mine = torch.tensor([dt / args.steps * 1e3, rank_kernel_ms[0], float(eng.num_local_parts), setup_s], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(e[0].item()) for e in every]
        rank_kernel_ms = [float(e[1].item()) for e in every]
        rank_parts = [int(e[2].item()) for e in every]
        rank_setup_s = [float(e[3].item()) for e in every]
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_parts = eng.total_parts
    local_parts = eng.num_local_parts
    moves_timed_all = total_parts * args.moves_per_part * args.steps
    value = moves_timed_all / dt
    log_G, log_prior = eng.global_totals()
    check_scale = None
    if world > 1 and args.verify_parts > 0:
        eng.backend_device = local_rank
        check_scale = scale_check(args, sc, eng, args.warmup + args.steps + len(ev_ms), world, rank, shared_gpu, dist, torch)

    # roofline of the dominant kernel (k_run_moves): algorithmic bytes counted by the kernel per executed move
    launches = args.steps + len(ev_ms)
    # (of the parts that ran in that kernel: the few parts of the side classes run beside it in k_run_moves_side)
    in_main = eng.backend.main_class_mask(eng.num_local_parts)
    per_part = np.array(stats1["algorithmic_bytes_of_part"], np.float64) - np.array(stats0["algorithmic_bytes_of_part"], np.float64)
    per_part_w = np.array(stats1["algorithmic_write_bytes_of_part"], np.float64) - np.array(stats0["algorithmic_write_bytes_of_part"], np.float64)
    write_bytes_per_launch = float(per_part_w[in_main].sum()) / max(1, launches)
    bytes_per_launch = float(per_part[in_main].sum()) / max(1, launches)
    bytes_per_launch_all = float(per_part.sum()) / max(1, launches)
    avg_ms = float(np.mean(ev_ms)) if ev_ms else float("nan")
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, traffic_source, traffic_read, traffic_write = None, None, None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    # PMC counters cannot be read from inside this process: the figure comes from the committed rocprofv3 --pmc passes
    # of this same command (scripts/profile.sh) and is stamped with the kernel build it was measured on; it is quoted
    # only for the default single-GPU workload and only while the device code is the one that was profiled
    if world == 1 and args.workload == "C4" and args.tips is None and args.parts == 8192 and args.moves_per_part == 1000 and args.max_part_nodes == -1 and not (args.no_topology or args.only_displace or args.no_lds):
        try:
            pm = json.load(open(pmc_path))
            stamp = pm.get("device_code_sha16")
            if stamp is None:
                traffic_source = "unstamped: profiles/pmc_latest.json does not say which device code it was measured on"
            elif stamp == build_id:
                traffic = pm.get("hbm_bytes_per_launch")
                # reads and writes apart: their sum can sit near the algorithmic bytes while each is far from its own share
                # (the state lives in LDS: few reads; private frames of the topology moves: many writes)
                if pm.get("FETCH_SIZE_KiB") is not None and pm.get("WRITE_SIZE_KiB") is not None:
                    traffic_read, traffic_write = 2.0 * pm["FETCH_SIZE_KiB"] * 1024.0, pm["WRITE_SIZE_KiB"] * 1024.0
                traffic_source = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on the library with emat_build_id %s)" % (pm.get("profile_dir", "pmc_latest.json"), stamp)
            else:
                traffic_source = "stale: profiles/pmc_latest.json was measured on device code %s, the loaded library is %s" % (stamp, build_id)
        except Exception as e:
            traffic, traffic_source = None, "unreadable profiles/pmc_latest.json: %s" % e
    bad = stats1["bad_parts"]
    # what a move buys: effective samples per second at the benchmark's part density next to the reference's few-part policy, from the
    # committed posterior-equivalence run (scripts/posterior_check.py; a 500-tip tree, so its moves/s are not this workload's)
    mixing = None
    mix_path = os.path.join(ROOT, "profiles", "posterior_latest.json")
    if rank == 0 and os.path.exists(mix_path):
        try:
            pc = json.load(open(mix_path))
            if pc.get("emat_build_id") != build_id:
                raise KeyError("measured on device code %s, the loaded library is %s: not quoted (re-run scripts/posterior_check.py)" % (pc.get("emat_build_id"), build_id))
            mixing = {"measured_on_emat_build_id": pc.get("emat_build_id"),
                      "source": "profiles/posterior_latest.json (scripts/posterior_check.py: %d tips, %d retained samples per arm, one per cycle of 50 x nodes moves)" % (pc["tips"], pc["configs"][0]["retained"]),
                      "worst_abs_z_between_arms": pc.get("worst_abs_z"),
                      "arms": [{"parts": c["parts"], "frozen_fraction": c["frozen_fraction"], "moves_per_s": c["moves_per_s"],
                                "ess_per_s": {k: v["ess_per_s"] for k, v in c["stats"].items()},
                                "ess_per_million_moves": {k: v["ess_per_million_moves"] for k, v in c["stats"].items()}} for c in pc["configs"]]}
        except Exception as e:
            mixing = {"omitted": "profiles/posterior_latest.json: %s" % e}
    # ... and at a size nearer the benchmark's (tests/posterior_scale.py: C3-like tree, 4 seeds; the coarse arm is the oracle with the
    # reference's policy of 8 parts on 8 host threads, the fine arms run on the GPU): pooled z of the arms' means and ESS per second
    scale_path = os.path.join(ROOT, "profiles", "posterior_scale_latest.json")
    if rank == 0 and os.path.exists(scale_path):
        try:
            ps = json.load(open(scale_path))
            if ps.get("emat_build_id") != build_id:
                raise KeyError("measured on device code %s, the loaded library is %s: not quoted (re-run tests/posterior_scale.py)" % (ps.get("emat_build_id"), build_id))
            mixing = mixing or {}
            mixing["at_scale"] = {"source": "profiles/posterior_scale_latest.json (tests/posterior_scale.py: %d tips, %d seeds, burn-in %d cycles)" % (ps["tips"], ps["seeds"], ps["burn_in"]),
                                  "measured_on_emat_build_id": ps.get("emat_build_id"), "part_size_limit": ps.get("part_size_limit"),
                                  "worst_abs_pooled_z": ps.get("worst_abs_pooled_z"),
                                  # (every run of the comparison made on this round's code, not only the last one, and what they say together)
                                  "replicates": ps.get("replicates"),
                                  "pooled_z": {arm: {k: v["pooled_z"] for k, v in q.items()} for arm, q in ps.get("pooled", {}).items()},
                                  "ess_per_s": {"gpu_at_benchmark_density": ps.get("ess_per_s_at_benchmark_density"), "oracle_8_parts_8_threads": ps.get("ess_per_s_reference_policy_oracle")}}
        except Exception as e:
            mixing = dict(mixing or {}, at_scale={"omitted": "profiles/posterior_scale_latest.json: %s" % e})

    # ... and the run driver's part-size limit against the reference's partition rule, device arms only (tests/posterior_limit.py)
    lim_path = os.path.join(ROOT, "profiles", "posterior_limit_latest.json")
    if rank == 0 and os.path.exists(lim_path):
        try:
            pl = json.load(open(lim_path))
            if pl.get("emat_build_id") != build_id:
                raise KeyError("measured on device code %s, the loaded library is %s: not quoted (re-run tests/posterior_limit.py)" % (pl.get("emat_build_id"), build_id))
            mixing = mixing or {}
            mixing["part_size_limit"] = {"source": "profiles/posterior_limit_latest.json (tests/posterior_limit.py: %d tips, %d parts requested, %d seeds, %d cycles, burn-in %d; every arm on the GPU, "
                                                   "the arm under the reference's rule against the arms with the limit)" % (pl["tips"], pl["parts_requested"], pl["seeds"], pl["cycles"], pl["burn_in"]),
                                         "measured_on_emat_build_id": pl.get("emat_build_id"), "worst_abs_pooled_z": pl.get("worst_abs_pooled_z"),
                                         "arms": {arm: {"limit_in_effect": q.get("limit_in_effect"), "cut_nodes_added_per_cycle": q.get("cut_nodes_added_per_cycle"),
                                                        "pooled_z": {k: v["pooled_z"] for k, v in q.items() if isinstance(v, dict)}} for arm, q in pl.get("pooled", {}).items()}}
        except Exception as e:
            mixing = dict(mixing or {}, part_size_limit={"omitted": "profiles/posterior_limit_latest.json: %s" % e})

    # The same measurement on the workload that HAS the parts to fill eight GPUs (C5: 1 000 000 tips, about 80 000 parts -- the 100 000-tip
    # tree of the metric yields about 8 000, fewer than one GPU has wave slots from N = 2 on), so that a scaling run shows both curves.
    secondary = None
    if args.secondary and args.secondary != args.workload and args.tips is None:
        secondary = measure_secondary(args, world, rank, local_rank, shared_gpu, allreduce, dist, torch)

    # Both decompositions side by side: `value` at N = 1 runs the 8192-part request, at N >= 2 the request grows with the GPUs and the moves per part
    # shrink so that a step keeps its moves.  A scaling curve read off `value` alone would mix the two: N = 1 therefore also times the grown partition, and N >= 2 the
    # fixed one, through the same timed region.
    decompositions = None
    if args.parts_auto and args.moves_auto and args.workload == "C4" and args.tips is None and not args.no_decompositions:
        step_moves = 1000.0 * base_parts
        if world == 1:
            decompositions = {"grown_partition": dict(measure_resident(args, sc, 8 * base_parts, 1000, args.steps, world, rank, local_rank, shared_gpu, allreduce, dist, torch, moves_of_a_step=step_moves),
                                                      what="the partition bench.py --gpus N >= 2 runs (8192 parts requested per GPU; the partitioner's floor of ten nodes per part caps C4 at about 13 000), on this one GPU, "
                                                           "with the moves of a step kept: the x 1 of a scaling curve read on that decomposition")}
        else:
            decompositions = {"fixed_partition": dict(measure_resident(args, sc, base_parts, 1000, args.steps, world, rank, local_rank, shared_gpu, allreduce, dist, torch),
                                                      what="N = 1's partition (8192 parts requested, 1000 moves per part) sharded over these ranks: the other decomposition, on which a rank of N >= 2 holds "
                                                           "fewer parts than wave slots and its pass is its slowest chain"),
                              "grown_partition": {"value": value, "ms_per_step": dt / args.steps * 1e3, "parts": total_parts, "parts_requested": args.parts, "moves_per_part": args.moves_per_part,
                                                  "what": "the headline `value` of this line"}}

    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(sc, args.parts, 20261001, args.cpu_seconds, eng.t_step)
    inclusive = None
    if rank == 0 and world == 1 and not args.no_inclusive:
        inclusive = inclusive_cycles(sc, args, args.inclusive_cycles)
    if world > 1 and not args.no_inclusive:
        # whole sharded cycles: the reference's steps/s is inclusive of everything (tools/delphy.cpp:44-51; run.cpp:622-657), and so is this -- on both decompositions.
        # (The headline above is measured and stays whatever happens here: a failure of this section -- it has only ever run with all ranks on one GPU -- is reported
        # in the line instead of taking the line down; a failure that is not the same on every rank would still hang the collectives, which the launcher's timeout ends.)
        cyc = max(4, min(args.inclusive_cycles, 60))
        try:
            inclusive = inclusive_sharded(sc, args, args.parts, cyc, world, rank, local_rank, shared_gpu, dist, torch)
            if args.parts != base_parts:
                inclusive["fixed_partition"] = inclusive_sharded(sc, args, base_parts, max(4, cyc // 3), world, rank, local_rank, shared_gpu, dist, torch)
        except Exception as e:      # noqa: BLE001
            inclusive = dict(inclusive or {}, error="%s: %s" % (type(e).__name__, e))

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
                "parts_per_gpu": local_parts, "parts_requested": args.parts, "max_part_nodes": args.max_part_nodes,
                "lds_staging": not args.no_lds,
                "tickets_per_part_and_pass": int(os.environ.get("EMAT_CHUNKS", "4")),
                "parallelism": "parts sharded over %d GPU(s), one wavefront per part" % world,
                "collectives": ("gloo on one shared GPU (plumbing check)" if shared_gpu else "RCCL, world size %d" % dist.get_world_size()) if world > 1 else "none (one rank)",
                "emat_build_id": build_id,
            },
            "per_rank": {"ms_per_step": rank_ms, "kernel_ms": rank_kernel_ms, "parts": rank_parts, "setup_s": rank_setup_s},
            "secondary": secondary,
            "decompositions": decompositions,
            # "bound" names the yardstick the contract asks for; "limiter" says what actually limits the kernel
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_read": traffic_read, "traffic_write": traffic_write, "traffic_source": traffic_source,
                         "limiter": "instruction issue and LDS / L2 latency: a part's chain is serial and runs on one lane of its wavefront (the wave's other lanes "
                                    "take part in slab staging and in the candidate scan + study of SPR moves); residency is capped by LDS at 16 parts per CU",
                         "kernel": "k_run_moves", "kernel_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         # reads and writes apart, each with its own denominator: what the moves must read / must store, counted by the kernel
                         "algorithmic_read_bytes": bytes_per_launch - write_bytes_per_launch, "algorithmic_write_bytes": write_bytes_per_launch,
                         "traffic_read_over_algorithmic": (traffic_read / (bytes_per_launch - write_bytes_per_launch)) if traffic_read else None,
                         "traffic_write_over_algorithmic": (traffic_write / write_bytes_per_launch) if (traffic_write and write_bytes_per_launch > 0) else None,
                         "parts_in_kernel": int(in_main.sum()), "algorithmic_bytes_per_pass_all_parts": bytes_per_launch_all},
            "cpu_baseline": cpu_base,
            "inclusive": inclusive,
            "mixing": mixing,
            "check": {"log_G": log_G, "log_augmented_coalescent_prior": log_prior, "parts_stopped": bad},
            "scale_check": check_scale,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
