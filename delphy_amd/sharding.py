"""Sharding of a partitioned EMAT over the GPUs of one node.

One process per GPU.  Every rank partitions the (deterministic, seeded) tree identically with the host
driver, keeps a contiguous block of the parts, and runs them on its own GPU.  Parts are independent between
`repartition` and `reassemble` (reference core/run.cpp:682-693), so the data path has no collective; the only
exchanges are the tiny all-reduces of the augmented coalescent grid when the parts are (re)built
(reference core/very_scalable_coalescent.cpp:153-219) and of the two log-posterior totals
(reference core/run.cpp:340-348).  `allreduce` is injected so that the same code runs over RCCL
(bench.py, backend "nccl") and over gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from .engine import EmatBackend, EmatRun
from .scenarios import Scenario


def block_range(num_items: int, rank: int, world: int):
    """Contiguous block of `num_items` owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _torch_allreduce(device: str):
    import torch
    import torch.distributed as dist

    def allreduce(arr: np.ndarray, op: str) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op])
        return t.cpu().numpy()
    return allreduce


class ShardedEngine:
    def __init__(self, sc: Scenario, num_parts: int, seed: int, rank: int = 0, world: int = 1, device: int = 0, use_lds: bool = True,
                 allreduce: Optional[Callable[[np.ndarray, str], np.ndarray]] = None, trace_moves: int = 0, t_step: Optional[float] = None,
                 max_part_nodes: int = 0):
        self.sc, self.num_parts_requested, self.seed, self.rank, self.world = sc, num_parts, seed, rank, world
        self.t_step = t_step if t_step is not None else sc.default_t_step()
        if allreduce is None:
            allreduce = (lambda a, op: a) if world == 1 else _torch_allreduce("cuda:%d" % device)
        self.allreduce = allreduce
        self.backend = EmatBackend(sc.num_sites, device=device, use_lds=use_lds, trace_moves=trace_moves)
        self.total_parts = 0
        self.num_local_parts = 0
        self.part_lo = self.part_hi = 0
        self.root_part = -1
        self.topology = True
        self.only_displace = False
        self.max_part_nodes = max_part_nodes   # not in the reference: cut larger parts further (0 = the reference's rule)

    def close(self):
        self.backend.close()

    def setup(self):
        sc = self.sc
        run = EmatRun(None, sc.tree, sc.ref, self.seed)   # host-only driver: same partition on every rank
        run.set_num_parts(self.num_parts_requested)
        run.set_max_part_nodes(self.max_part_nodes)
        run.repartition()
        n, root_part = run.num_parts()
        self.total_parts, self.root_part = n, root_part
        self.part_lo, self.part_hi = block_range(n, self.rank, self.world)
        parts, incl, seeds = [], [], []
        for i in range(self.part_lo, self.part_hi):
            t, r, s = run.part(i)
            parts.append(t); incl.append(r); seeds.append(s)
        _, ref = run.tree()
        run.close()
        self.num_local_parts = len(parts)
        self.local_sizes = [p.num_nodes for p in parts]
        b = self.backend
        b.set_ref_sequence(ref)
        b.set_hky(sc.mu, sc.kappa, sc.pi, sc.nu_l)
        b.set_flags(sc.t_max_tip, self.only_displace, self.topology)
        b.upload_parts(parts, incl, seeds)
        self.build_coalescent()

    def build_coalescent(self):
        """very_scalable_coalescent.cpp:85-232 with its three cross-part reductions done as all-reduces."""
        b = self.backend
        local_root = self.root_part - self.part_lo if self.part_lo <= self.root_part < self.part_hi else -1
        lo, hi = b.coalescent_begin(self.sc.pop, local_root, self.t_step)
        lo = float(self.allreduce(np.array([lo]), "min")[0])
        hi = float(self.allreduce(np.array([hi]), "max")[0])
        b.coalescent_set_range(lo, hi)
        k_bar, num_active = b.coalescent_local_grid()
        k_bar = self.allreduce(k_bar, "sum")
        num_active = self.allreduce(num_active.astype(np.int64), "sum").astype(np.int32)
        k_tw = b.coalescent_sample(k_bar, num_active)
        k_tw = self.allreduce(k_tw, "sum")
        b.coalescent_finish(k_tw)

    def local_stats(self):
        tot = dict(algorithmic_bytes=0, moves_done=0, bad_parts=0, proposed=[0] * 5, accepted=[0] * 5)
        for p in range(self.num_local_parts):
            s = self.backend.part_stats(p)
            tot["algorithmic_bytes"] += s["algorithmic_bytes"]
            tot["moves_done"] += s["moves_done"]
            tot["bad_parts"] += 1 if s["status"] != 0 else 0
            for k in range(5):
                tot["proposed"][k] += s["proposed"][k]
                tot["accepted"][k] += s["accepted"][k]
        return tot

    def global_stats(self, num_partitions: int = 1):
        """Sufficient statistics of the global moves over ALL parts: per-GPU sums, then one all-reduce each."""
        T, M, nm = self.backend.global_stats(num_partitions)
        T = self.allreduce(T.reshape(-1), "sum").reshape(num_partitions, 4)
        M = self.allreduce(M.reshape(-1).astype(np.int64), "sum").reshape(num_partitions, 4, 4)
        nm = int(self.allreduce(np.array([nm], np.int64), "sum")[0])
        return T, M, nm

    def global_totals(self):
        g, a = self.backend.totals()
        v = self.allreduce(np.array([g, a]), "sum")
        return float(v[0]), float(v[1])
