"""Ad-hoc GPU probe: long parity runs + first timings (not part of the test suite)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import run_parity, split_parts, configure

def timing(name, tips, sites, nparts, moves, use_lds=True):
    t0 = time.time()
    sc = make_scenario(name, num_tips=tips, num_sites=sites)
    t1 = time.time()
    parts, incl, seeds, root_part, ref = split_parts(sc, nparts, 5)
    t2 = time.time()
    gpu = d.EmatBackend(sc.num_sites, use_lds=use_lds)
    configure(gpu, sc, ref, parts, incl, seeds, root_part)
    gpu.recalc_derived(); gpu.synchronize()
    t3 = time.time()
    gpu.run_moves_per_part(moves); gpu.synchronize(); ms1 = gpu.last_run_ms()
    gpu.run_moves_per_part(moves); gpu.synchronize(); ms = gpu.last_run_ms()
    sizes = np.array([p.num_nodes for p in parts])
    bad = [gpu.part_stats(p)["status"] for p in range(len(parts))]
    nb = sum(1 for b in bad if b != 0)
    st = gpu.part_stats(0)
    print("%s tips=%d parts=%d (nodes/part min %d med %d max %d) lds=%s: gen %.1fs part %.1fs setup %.1fs | %d moves/part: %.2f ms (first %.2f) -> %.2f M moves/s | bad parts %d | alg bytes/move %.0f"
          % (name, tips, len(parts), sizes.min(), np.median(sizes), sizes.max(), use_lds, t1 - t0, t2 - t1, t3 - t2, moves, ms, ms1, len(parts) * moves / ms / 1e3, nb,
             st["algorithmic_bytes"] / max(1, st["moves_done"])), flush=True)
    if nb:
        print("   first bad:", [(i, b) for i, b in enumerate(bad) if b][:5], gpu.last_error())
    gpu.close()

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "long"):
        sc = make_scenario("C1", num_tips=100, num_sites=3000, uncertain_tips=0.3)
        t0 = time.time(); run_parity(sc, 1, 100000, trace=100000); print("long single root part 100k moves ok %.1fs" % (time.time() - t0), flush=True)
        sc = make_scenario("C2", num_tips=400, num_sites=4000, uncertain_tips=0.2)
        t0 = time.time(); run_parity(sc, 12, 50000, trace=50000); print("long C2 12 parts 50k moves/part ok %.1fs" % (time.time() - t0), flush=True)
        sc = make_scenario("C3", num_tips=2000, num_sites=29903)
        t0 = time.time(); run_parity(sc, 64, 20000, trace=20000); print("long C3-2k 64 parts 20k moves/part ok %.1fs" % (time.time() - t0), flush=True)
    if what in ("all", "time"):
        timing("C1", 100, 30000, 1, 20000)
        timing("C2", 1610, 18959, 64, 5000)
        timing("C3", 10000, 29903, 512, 2000)
        timing("C3", 10000, 29903, 512, 2000, use_lds=False)
        timing("C4", 100000, 29903, 4096, 1000)
        timing("C4", 100000, 29903, 8192, 1000)
        timing("C4", 100000, 29903, 8192, 1000, use_lds=False)


def tail_probe(nparts=8192, moves=1000):
    """Per-part device time distribution: which waves set the kernel's duration?"""
    from delphy_amd.sharding import ShardedEngine
    sc = make_scenario("C4")
    eng = ShardedEngine(sc, num_parts=nparts, seed=20261001)
    eng.setup()
    eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    t0 = [eng.backend.part_stats(p)["device_ticks"] for p in range(eng.num_local_parts)]
    eng.backend.run_moves_per_part(moves); eng.backend.synchronize(); ms = eng.backend.last_run_ms()
    st = [eng.backend.part_stats(p) for p in range(eng.num_local_parts)]
    ticks = np.array([s["device_ticks"] for s in st]) - np.array(t0)
    us = ticks / 100.0
    sizes = np.array(eng.local_sizes)
    print("kernel %.2f ms; per-part serial time (ms): min %.2f med %.2f p90 %.2f p99 %.2f max %.2f" % (ms, us.min() / 1e3, np.median(us) / 1e3, np.percentile(us, 90) / 1e3, np.percentile(us, 99) / 1e3, us.max() / 1e3))
    order = np.argsort(-us)[:12]
    for i in order:
        print("   part %5d nodes %4d root %s time %.2f ms" % (i, sizes[i], i == eng.root_part - eng.part_lo, us[i] / 1e3))
    print("   sum of per-part times %.1f ms-waves => avg concurrency %.0f waves" % (us.sum() / 1e3, us.sum() / 1e3 / ms))
    print("   part sizes (nodes): p50 %d p90 %d p95 %d p99 %d max %d" % tuple(np.percentile(sizes, [50, 90, 95, 99, 100])))
    c = np.corrcoef(sizes, us)[0, 1]
    print("   corr(size, time) = %.2f" % c)
    eng.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "tail":
    tail_probe()


def phase_probe(nparts=8192, moves=1000):
    """Needs a library built with -DEMAT_PROFILE_PHASES; reads the phase ticks straight out of the slab headers."""
    import ctypes as C
    from delphy_amd.sharding import ShardedEngine
    sc = make_scenario("C4")
    eng = ShardedEngine(sc, num_parts=nparts, seed=20261001)
    eng.setup()
    eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    lib = d.load_library()
    buf = (C.c_int64 * 16)()
    tot = np.zeros(16)
    lib.emat_debug_phase_ticks.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    root_local = eng.root_part - eng.part_lo
    for p in range(eng.num_local_parts):
        lib.emat_debug_phase_ticks(eng.backend.handle, p, buf)
        tot += np.array(list(buf), dtype=np.float64)
        if p == root_local:
            rp = np.array(list(buf), dtype=np.float64); st = eng.backend.part_stats(p)
            print("root part %d (%d nodes): simple-move ticks %.3g, topology-move ticks %.3g, proposed %s, device ms %.1f" % (p, st["num_nodes"], rp[14], rp[15], st["proposed"], st["device_ticks"] / 1e5))
    names = ["core:analyze+peel", "core:topology", "core:propose", "(unused)", "core:coal+accept+apply", "spr1:analyze+peel", "spr1:missing+seed_fill pre", "spr1:study pre+pick",
             "spr1:topology", "spr1:propose", "spr1:seed_fill post", "spr1:study post+alpha", "spr1:accept+apply", "regions (count)", "ALL simple moves", "ALL topology moves"]
    total = tot[14] + tot[15]
    for i, nme in enumerate(names):
        print("%-28s %14.0f  %5.1f%%" % (nme, tot[i], 100 * tot[i] / total if i != 13 else 0))
    st = eng.local_stats()
    print("proposed", st["proposed"], "regions per study %.1f" % (tot[13] / max(1, 2 * st["proposed"][4])))
    eng.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "phase":
    phase_probe()


def size_probe(nparts=8192, moves=1000):
    """Phase ticks per part-size bucket (needs the -DEMAT_PROFILE_PHASES library)."""
    import ctypes as C
    from delphy_amd.sharding import ShardedEngine
    sc = make_scenario("C4")
    world, rank = int(os.environ.get("EMAT_PROBE_WORLD", "1")), int(os.environ.get("EMAT_PROBE_RANK", "0"))   # one GPU of N, emulated (scripts/scale_probe.py)
    eng = ShardedEngine(sc, num_parts=nparts, seed=20261001, rank=rank, world=world, device_tree=world > 1, allreduce=lambda a, op: a, allgather_bytes=lambda b: [b])
    eng.setup()
    eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    lib = d.load_library()
    lib.emat_debug_phase_ticks.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    buf = (C.c_int64 * 16)()
    rows = []
    for p in range(eng.num_local_parts):
        lib.emat_debug_phase_ticks(eng.backend.handle, p, buf)
        st = eng.backend.part_stats(p)
        rows.append([eng.local_sizes[p]] + list(buf) + st["proposed"] + [st["device_ticks"]])
    a = np.array(rows, dtype=np.float64)
    edges = [0, 15, 20, 25, 30, 40, 50, 70, 1000]
    print("size-bucket  parts  simple_ticks/move  topo_ticks/topo_move  spr1_seedfill/spr1  regions/study  wall_ms")
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (a[:, 0] >= lo) & (a[:, 0] < hi)
        if not m.any():
            continue
        b = a[m]
        nsimple = b[:, 17:20].sum(); ntopo = b[:, 20:22].sum(); nspr1 = b[:, 21].sum()
        print("%4d-%-4d   %6d   %12.0f   %14.0f   %14.0f   %8.1f   %8.2f" % (lo, hi, m.sum(), b[:, 15].sum() / nsimple, b[:, 16].sum() / max(1, ntopo),
              (b[:, 7].sum() + b[:, 11].sum()) / max(1, nspr1), b[:, 14].sum() / max(1, 2 * nspr1), b[:, 22].mean() / 1e5))
    eng.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "size":
    size_probe()


def cycle_probe(nparts=8192, moves=1000, max_part_nodes=0):
    """One whole host cycle through the C++ run driver at C4: repartition (host partitioning + slab encode + H2D),
    local moves, reassemble (D2H + decode + host gather).  Gives the PCIe- and host-inclusive rate quoted in DESIGN.md."""
    import time
    sc = make_scenario("C4")
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, 20261001)
    run.set_num_parts(nparts)
    run.set_max_part_nodes(max_part_nodes)
    run.set_hky(sc.mu, sc.kappa, sc.pi)
    run.set_pop_model(sc.pop)
    for cyc in range(4):
        t0 = time.perf_counter(); run.repartition(); n, _ = run.num_parts(); t1 = time.perf_counter()   # repartition pushes the model and builds the coalescent parts
        run.run_moves(n * moves); b.synchronize(); t2 = time.perf_counter()
        ms = b.last_run_ms()
        run.reassemble(); t3 = time.perf_counter()
        st = [b.part_stats(p) for p in range(n)]
        d_t = np.array([x["device_ticks"] for x in st], dtype=np.float64)   # parts are re-uploaded every cycle: ticks restart
        top = np.argsort(-d_t)[:4]
        _, rp = run.num_parts()
        slowest = "   slowest parts: " + ", ".join("part %d nodes %d %.1f ms" % (i, st[i]["num_nodes"], d_t[i] / 1e5) for i in top) + " | root part %d: %.1f ms" % (rp, d_t[rp] / 1e5)
        print("cycle %d: %d parts | repartition+upload %.1f ms | moves %.1f ms (kernel %.1f ms; first call includes slab build + H2D + recalc) | reassemble (D2H + gather) %.1f ms | "
              "whole cycle %.1f ms => %.1f M moves/s inclusive vs %.1f M moves/s resident" % (cyc, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ms, (t3 - t2) * 1e3, (t3 - t0) * 1e3,
              n * moves / (t3 - t0) / 1e6, n * moves / (ms * 1e-3) / 1e6))
        print(slowest)
    run.close(); b.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "cycle":
    cycle_probe(max_part_nodes=int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def timeline_probe(nparts=8192, moves=1000):
    """Occupancy timeline of one pass: how many parts are running at each instant (from per-part start ticks and durations)."""
    import ctypes as C
    from delphy_amd.sharding import ShardedEngine
    sc = make_scenario("C4")
    eng = ShardedEngine(sc, num_parts=nparts, seed=20261001)
    eng.setup()
    for _ in range(2):
        eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    ms = eng.backend.last_run_ms()
    n = eng.num_local_parts
    buf = (C.c_int64 * (2 * n))()
    lib = d.load_library()
    lib.emat_debug_part_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    assert lib.emat_debug_part_ticks(eng.backend.handle, buf) == 0
    a = np.array(list(buf), dtype=np.float64)
    dur, start = a[:n] / 1e5, a[n:] / 1e5   # ms
    start -= start.min(); end = start + dur
    print("kernel %.2f ms; parts %d; last end %.2f ms; sum %.0f ms-waves" % (ms, n, end.max(), dur.sum()))
    edges = np.linspace(0, end.max(), 25)
    for lo, hi in zip(edges[:-1], edges[1:]):
        mid = 0.5 * (lo + hi)
        running = int(((start <= mid) & (end > mid)).sum()); started = int(((start >= lo) & (start < hi)).sum())
        print("  t=%5.1f ms  running %5d  started in bin %5d" % (mid, running, started))
    sizes = np.array(eng.local_sizes)
    late = np.argsort(-end)[:8]
    for i in late: print("   part %5d nodes %4d start %.1f dur %.1f end %.1f" % (i, sizes[i], start[i], dur[i], end[i]))
    eng.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "timeline":
    timeline_probe()


def parts_probe(nparts=8192, moves=1000, out_csv="gpurun_out/parts_probe.csv"):
    """Per-part features next to the part's duration in one pass: what makes a chain slow?"""
    import ctypes as C
    sc = make_scenario("C4")
    parts, incl, seeds, root_part, ref = split_parts(sc, nparts, 20261001)
    gpu = d.EmatBackend(sc.num_sites)
    configure(gpu, sc, ref, parts, incl, seeds, root_part)
    gpu.run_moves_per_part(moves); gpu.synchronize()
    t0 = np.array([gpu.part_stats(p)["device_ticks"] for p in range(len(parts))], dtype=np.float64)
    p0 = np.array([gpu.part_stats(p)["proposed"] for p in range(len(parts))], dtype=np.float64)
    gpu.run_moves_per_part(moves); gpu.synchronize(); ms = gpu.last_run_ms()
    st = [gpu.part_stats(p) for p in range(len(parts))]
    dur = (np.array([s["device_ticks"] for s in st], dtype=np.float64) - t0) / 1e5
    prop = np.array([s["proposed"] for s in st], dtype=np.float64) - p0
    acc = np.array([s["accepted"] for s in st], dtype=np.float64)
    n = len(parts)
    buf = (C.c_int64 * (2 * n))()
    lib = d.load_library()
    lib.emat_debug_part_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    assert lib.emat_debug_part_ticks(gpu.handle, buf) == 0
    a = np.array(list(buf), dtype=np.float64); start = a[n:] / 1e5; start -= start.min()
    feats = np.array([[p.num_nodes, p.mut_site.shape[0], p.miss_start.shape[0], p.mfs_site.shape[0], int(p.mut_offset[1] - p.mut_offset[0]),
                       16 * p.mut_site.shape[0] + 8 * p.miss_start.shape[0] + 8 * p.mfs_site.shape[0] + 64 * p.num_nodes] for p in parts], dtype=np.float64)
    # depth statistics
    depth = []
    for p in parts:
        dp = np.zeros(p.num_nodes, np.int32)
        order = np.argsort(p.t, kind="stable")
        for i in order:
            if p.parent[i] >= 0: dp[i] = dp[p.parent[i]] + 1
        depth.append([dp.mean(), dp.max()])
    depth = np.array(depth)
    M = np.column_stack([feats, depth, dur, start, prop, acc[:, 3:5]])
    hdr = "nodes,muts,ivs,fss,root_muts,bytes,depth_mean,depth_max,dur_ms,start_ms,p_inner,p_tip,p_reform,p_slide,p_spr1,a_slide,a_spr1"
    np.savetxt(os.path.join(ROOT, out_csv), M, delimiter=",", header=hdr, fmt="%.4g")
    print("kernel %.2f ms, parts %d" % (ms, n))
    names = hdr.split(",")
    for j, nm in enumerate(names):
        if nm != "dur_ms": print("  corr(dur, %-10s) = %+.2f" % (nm, np.corrcoef(M[:, j], dur)[0, 1]))
    # linear fit
    X = np.column_stack([np.ones(n), feats[:, 0], feats[:, 1], feats[:, 2], feats[:, 4], depth[:, 0]])
    coef, *_ = np.linalg.lstsq(X, dur, rcond=None)
    print("  dur ~ %.2f + %.3f nodes + %.4f muts + %.4f ivs + %.4f root_muts + %.3f depth_mean; resid std %.2f ms" % (*coef, np.std(dur - X @ coef)))
    for lo, hi in [(0, 9000), (9000, 9700), (9700, 12000), (12000, 16000), (16000, 10**9)]:
        m = (feats[:, 5] >= lo) & (feats[:, 5] < hi)
        if m.any(): print("  content bytes %6d-%-8d parts %5d dur mean %.2f p90 %.2f max %.2f ms" % (lo, hi, m.sum(), dur[m].mean(), np.percentile(dur[m], 90), dur[m].max()))
    gpu.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "parts":
    parts_probe()


def slow_phase_probe(nparts=8192, moves=1000, top=6):
    """Phase breakdown (needs the -DEMAT_PROFILE_PHASES library, EMAT_LIB_PATH) of the slowest parts of a pass."""
    import ctypes as C
    from delphy_amd.sharding import ShardedEngine
    sc = make_scenario("C4")
    world, rank = int(os.environ.get("EMAT_PROBE_WORLD", "1")), int(os.environ.get("EMAT_PROBE_RANK", "0"))   # one GPU of N, emulated (scripts/scale_probe.py)
    eng = ShardedEngine(sc, num_parts=nparts, seed=20261001, rank=rank, world=world, device_tree=world > 1, allreduce=lambda a, op: a, allgather_bytes=lambda b: [b])
    eng.setup()
    eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    lib = d.load_library()
    lib.emat_debug_phase_ticks.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    buf = (C.c_int64 * 16)()
    st = [eng.backend.part_stats(p) for p in range(eng.num_local_parts)]
    dur = np.array([s["device_ticks"] for s in st], dtype=np.float64) / 1e5
    names = ["core:analyze+peel", "core:topology", "core:propose", "scans in LDS", "core:coal+accept+apply", "spr1:analyze+peel", "spr1:missing+seed_fill pre", "spr1:study pre+pick",
             "spr1:topology", "spr1:propose", "spr1:seed_fill post", "spr1:study post+alpha", "spr1:accept+apply", "regions (count)", "ALL simple moves", "ALL topology moves"]
    sel = list(np.argsort(-dur)[:top]) + list(np.argsort(dur)[len(dur) // 2: len(dur) // 2 + 2])
    for p in sel:
        lib.emat_debug_phase_ticks(eng.backend.handle, int(p), buf)
        v = np.array(list(buf), dtype=np.float64)
        tot = v[14] + v[15]
        print("part %d nodes %d dur %.2f ms proposed %s | regions/study %.1f" % (p, eng.local_sizes[p], dur[p], st[p]["proposed"], v[13] / max(1, 2 * st[p]["proposed"][4])))
        print("    " + " | ".join("%s %.1f%%" % (names[i], 100 * v[i] / tot) for i in list(range(0, 3)) + list(range(4, 13)) + [14, 15]))
        os.environ["EMAT_PHASE_EXTRA"] = "1"
        lib.emat_debug_phase_ticks(eng.backend.handle, int(p), buf)
        del os.environ["EMAT_PHASE_EXTRA"]
        e = np.array(list(buf), dtype=np.float64); k = max(1.0, e[0])
        print("    wave scans %d: missing intervals at X %.1f, site deltas %.1f, items %.1f, levels %.1f, in HBM %d, fell back to serial %d, sets not in LDS %d | scan ticks per scan %.0f"
              % (e[0], e[1] / k, e[2] / k, e[3] / k, e[4] / k, e[5], e[6], e[7], v[6] / k))
        pr = st[p]["proposed"]
        print("    ticks per move: inner-node displacement %.0f, tip displacement %.0f, branch reform %.0f, subtree slide + SPR1 %.0f | cells per coalescent delta %.1f (%.2f deltas per move)"
              % (e[10] / max(1, pr[0]), e[15] / max(1, pr[1]), (v[14] - e[10] - e[15]) / max(1, pr[2]), v[15] / max(1, pr[3] + pr[4]), e[13] / max(1.0, e[14]), e[14] / max(1, sum(pr))))
        topo = max(1, st[p]["proposed"][3] + st[p]["proposed"][4])
        print("    arena bytes per topology move: %.0f from HBM scratch, %.0f from LDS (open vectors trimmed in HBM %d, spans committed in HBM %d)" % (e[8] / topo, e[9] / topo, e[11], e[12]))
    eng.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "slowphase":
    slow_phase_probe()


def fn_probe(nparts=8192, moves=1000):
    """Inclusive time per instrumented device function (EMAT_TIMED scopes of the -DEMAT_PROFILE_PHASES library, EMAT_LIB_PATH),
    summed over all parts of a pass, next to the time of all simple and all topology moves."""
    import ctypes as C
    from delphy_amd.sharding import ShardedEngine
    sc = make_scenario("C4")
    world, rank = int(os.environ.get("EMAT_PROBE_WORLD", "1")), int(os.environ.get("EMAT_PROBE_RANK", "0"))
    eng = ShardedEngine(sc, num_parts=nparts, seed=20261001, rank=rank, world=world, device_tree=world > 1, allreduce=lambda a, op: a, allgather_bytes=lambda b: [b])
    eng.setup()
    lib = d.load_library()
    lib.emat_debug_fn_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.emat_debug_phase_ticks.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    buf = (C.c_uint64 * 12288)()
    eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    assert lib.emat_debug_fn_ticks(eng.backend.handle, buf) == 0   # clears
    ph0 = np.zeros((eng.num_local_parts, 16)); pb = (C.c_int64 * 16)()
    for p in range(eng.num_local_parts):
        lib.emat_debug_phase_ticks(eng.backend.handle, p, pb); ph0[p] = list(pb)
    eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    assert lib.emat_debug_fn_ticks(eng.backend.handle, buf) == 0
    ph1 = np.zeros((eng.num_local_parts, 16))
    for p in range(eng.num_local_parts):
        lib.emat_debug_phase_ticks(eng.backend.handle, p, pb); ph1[p] = list(pb)
    min_lists = int(os.environ.get("EMAT_FN_MIN_LISTS", "0"))    # only the parts whose lists take at least this many bytes (the library filters its function timers alike)
    sel = np.array([eng.backend.debug_slab_layout(p)["heap_used"] >= min_lists for p in range(eng.num_local_parts)])
    print("parts counted: %d of %d (lists >= %d bytes)" % (sel.sum(), len(sel), min_lists))
    dph = (ph1 - ph0)[sel].sum(axis=0)
    simple, topo = dph[14], dph[15]
    a = np.array(list(buf), dtype=np.float64).reshape(3 * 2048, 2)
    files = ["emat_device_core.hpp", "emat_device_spr.hpp", "emat_device_moves.hpp"]
    src = [open(os.path.join(ROOT, "delphy_amd", "csrc", f)).read().split("\n") for f in files]
    rows = []
    for k in np.nonzero(a[:, 1])[0]:
        f, line = divmod(int(k), 2048)
        text = src[f][line - 1] if line - 1 < len(src[f]) else "?"
        import re
        m = re.search(r"(\w+)\s*\([^()]*(\([^()]*\))?[^()]*\)\s*\{ EMAT_TIMED", text)
        rows.append((a[k, 0], a[k, 1], (m.group(1) if m else text[:40]) + " (%s:%d)" % (files[f].replace("emat_device_", ""), line)))
    rows.sort(reverse=True)
    dticks = sum(eng.backend.part_stats(p)["device_ticks"] for p in range(eng.num_local_parts))
    print("pass %.2f ms; chains %.3g ticks of 10 ns over both passes" % (eng.backend.last_run_ms(), dticks))
    print("parts %d, moves %d: simple moves %.3g ticks, topology moves %.3g ticks (%.1f%% of chain time)" % (eng.num_local_parts, moves, simple, topo, 100 * topo / (simple + topo)))
    ntopo = sum(sum(eng.backend.part_stats(p)["proposed"][3:5]) for p in range(eng.num_local_parts) if sel[p]) / 2.0
    os.environ["EMAT_PHASE_EXTRA"] = "1"
    ex = np.zeros(16)
    for p in range(eng.num_local_parts):
        lib.emat_debug_phase_ticks(eng.backend.handle, p, pb); ex += np.array(list(pb), dtype=np.float64)
    del os.environ["EMAT_PHASE_EXTRA"]
    print("coalescent delta: %.0f calls over both passes, %.2f cells per call" % (ex[14], ex[13] / max(1.0, ex[14])))
    print("%-62s %10s %12s %10s %8s" % ("function (inclusive)", "% of topo", "calls/tmove", "ticks/call", "% chain"))
    for t, n, name in rows:
        print("%-62s %9.1f%% %12.2f %10.0f %7.1f%%" % (name, 100 * t / topo, n / ntopo, t / n, 100 * t / (simple + topo)))
    eng.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "fn":
    fn_probe()


def arena_probe(nparts=8192, moves=1000):
    """Where the scratch of the topology moves comes from, over ALL parts of a pass (-DEMAT_PROFILE_PHASES library): bytes taken
    from the LDS arena and from the part's HBM scratch region per topology move, and the allocation sites that went to HBM."""
    import ctypes as C
    from delphy_amd.sharding import ShardedEngine
    sc = make_scenario("C4")
    eng = ShardedEngine(sc, num_parts=nparts, seed=20261001, rank=0, world=1, device_tree=False, allreduce=lambda a, op: a, allgather_bytes=lambda b: [b])
    eng.setup()
    eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    lib = d.load_library()
    lib.emat_debug_phase_ticks.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    lib.emat_debug_arena_sites.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    buf = (C.c_int64 * 16)(); tot = np.zeros(16); topo = 0
    os.environ["EMAT_PHASE_EXTRA"] = "1"
    per_part = []
    for p in range(eng.num_local_parts):
        lib.emat_debug_phase_ticks(eng.backend.handle, p, buf); e = np.array(list(buf), dtype=np.float64); tot += e
        st = eng.backend.part_stats(p); t = st["proposed"][3] + st["proposed"][4]; topo += t
        per_part.append((e[8], e[9], t))
    del os.environ["EMAT_PHASE_EXTRA"]
    print("topology moves %d | arena bytes per topology move: %.0f from HBM scratch, %.0f from LDS | open vectors trimmed in HBM %d, spans committed in HBM %d"
          % (topo, tot[8] / topo, tot[9] / topo, tot[11], tot[12]))
    pp = np.array(per_part); frac = (pp[:, 0] > 0).mean()
    print("parts with any HBM scratch: %.1f%%; share of HBM bytes in the top 1%% of parts: %.1f%%" % (100 * frac, 100 * np.sort(pp[:, 0])[-max(1, len(pp) // 100):].sum() / max(1.0, pp[:, 0].sum())))
    sites = (C.c_uint64 * 4096)()
    if lib.emat_debug_arena_sites(eng.backend.handle, sites) == 0:
        a = np.array(list(sites), dtype=np.float64).reshape(2048, 2)
        order = np.argsort(-a[:, 1])[:12]
        print("allocation sites by HBM bytes (source line & 2047: HBM bytes per topology move, LDS bytes per topology move)")
        for k in order:
            if a[k, 0] + a[k, 1] > 0: print("   line %4d: %8.0f %8.0f" % (k, a[k, 1] / topo, a[k, 0] / topo))
    eng.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "arena":
    arena_probe()


def ticket_timeline_probe(nparts=8192, moves=1000):
    """Occupancy of a pass WITH tickets: resident workgroups of the move kernels over time, from the entry and exit ticks of every
    ticket (emat_debug_ticket_ticks), next to the slot time the chains themselves account for."""
    import ctypes as C
    from delphy_amd.sharding import ShardedEngine
    sc = make_scenario("C4")
    eng = ShardedEngine(sc, num_parts=nparts, seed=20261001)
    eng.setup()
    for _ in range(2):
        eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    ms = eng.backend.last_run_ms(); n = eng.num_local_parts
    lib = d.load_library()
    lib.emat_debug_ticket_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.emat_debug_part_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    buf = (C.c_int64 * (16 * n))(); assert lib.emat_debug_ticket_ticks(eng.backend.handle, buf) == 0
    pt = (C.c_int64 * (2 * n))(); assert lib.emat_debug_part_ticks(eng.backend.handle, pt) == 0
    a = np.array(list(buf), dtype=np.float64).reshape(8, 2, n) / 1e5   # ms
    chain = np.array(list(pt), dtype=np.float64)[:n] / 1e5
    used = a[:, 1, :] > 0
    t0 = a[:, 0, :][used].min()
    ent, ext = a[:, 0, :][used] - t0, a[:, 1, :][used] - t0
    print("pass %.2f ms; parts %d; tickets logged %d; last exit %.2f ms; resident slot time %.1f s, of which chains %.1f s" % (ms, n, int(used.sum()), ext.max(), (ext - ent).sum() / 1e3, chain.sum() / 1e3))
    per_ticket = [(a[k, 1, :] - a[k, 0, :])[used[k]] for k in range(8) if used[k].any()]
    print("ticket durations (ms), mean per ticket index:", ["%.2f" % x.mean() for x in per_ticket])
    flat = [(a[k, 1, p] - t0, k, p) for k in range(8) for p in range(n) if used[k, p]]
    flat.sort(reverse=True)
    sizes = np.array(eng.local_sizes)
    for e, k, p in flat[:8]:
        print("   last exits: part %5d (%d nodes%s) ticket %d entered %.2f exited %.2f ms; the part's chain ticks %.2f ms over all its tickets" % (p, sizes[p], ", root part" if p == eng.root_part - eng.part_lo else "", k, a[k, 0, p] - t0, e, chain[p]))
    edges = np.linspace(0, ext.max(), 41)
    for lo, hi in zip(edges[:-1], edges[1:]):
        mid = 0.5 * (lo + hi)
        print("  t=%5.2f ms  resident %5d  entered in bin %5d" % (mid, int(((ent <= mid) & (ext > mid)).sum()), int(((ent >= lo) & (ent < hi)).sum())))
    eng.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "tickets":
    ticket_timeline_probe()
