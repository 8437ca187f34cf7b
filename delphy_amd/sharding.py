"""One EMAT run sharded over the GPUs of a node: one process per GPU.

Every rank holds the whole tree and the same host driver state (include/emat_host.h), cuts the tree identically, and keeps
a contiguous block of the parts on its own GPU.  Between `repartition` and `reassemble` the parts are independent
(reference core/run.cpp:682-693), so the data path has no collective.  Per cycle the ranks exchange:

  * the three tiny reductions of the augmented coalescent grid when the parts are (re)built
    (reference core/very_scalable_coalescent.cpp:153-219): MIN / MAX of the time range, SUM of k_bar, SUM of k_twiddle_bar;
  * the parts themselves on the way back (reference Run::reassemble needs every part's tree, run.cpp:195-256): one
    all-gather of the serialised local parts, after which every rank reassembles the same whole tree;
  * SUM of the two log-posterior totals (reference run.cpp:340-348).

All of the run logic is behind the C-ABI (emat_run_set_shard, _coalescent_begin, _moves_sharded, _pack_local_parts,
_unpack_parts, _reassemble); this module only owns the collectives, which are injected (`allreduce`, `allgather_bytes`)
so that the same code runs over RCCL (bench.py: backend "nccl") and over gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np

from .engine import EmatBackend, EmatRun
from .scenarios import Scenario


def block_range(num_items: int, rank: int, world: int):
    """Contiguous block of `num_items` owned by `rank` (sizes differ by at most one); the rule of emat_run_set_shard."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _torch_collectives(device: str):
    import torch
    import torch.distributed as dist

    def allreduce(arr: np.ndarray, op: str) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op])
        return t.cpu().numpy()

    def allgather_bytes(buf: np.ndarray) -> List[np.ndarray]:
        world = dist.get_world_size()
        sizes = torch.tensor([buf.shape[0]], dtype=torch.int64, device=device)
        all_sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(all_sizes, sizes)
        n = [int(s.item()) for s in all_sizes]
        padded = torch.zeros(max(n), dtype=torch.uint8, device=device)
        padded[: buf.shape[0]] = torch.from_numpy(buf).to(device)
        out = [torch.zeros(max(n), dtype=torch.uint8, device=device) for _ in range(world)]
        dist.all_gather(out, padded)
        return [o[: n[r]].cpu().numpy() for r, o in enumerate(out)]
    return allreduce, allgather_bytes


def _torch_device_allgather(device: str):
    """All-gather of variable-length byte buffers that STAY on the device: `fill(tensor)` writes this rank's contribution
    into a device tensor of `nbytes`; returns one device tensor per rank (RCCL all-gather of the padded buffers)."""
    import torch
    import torch.distributed as dist

    def allgather_device(nbytes: int, fill):
        world = dist.get_world_size()
        sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([nbytes], dtype=torch.int64, device=device))
        n = [int(x.item()) for x in sizes]
        mine = torch.empty(max(max(n), 1), dtype=torch.uint8, device=device)
        # The engine's kernels run on the engine's own HIP stream, the collective on torch's current stream: neither orders
        # itself after the other.  `fill` ends with the engine's stream synchronised (emat_tree_export_nodes waits for its
        # kernels); the waits below close the other two gaps whatever stream the caller made current.
        torch.cuda.current_stream(device).synchronize()      # `mine` is allocated and nobody else is writing it
        fill(mine)
        out = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(out, mine)
        torch.cuda.current_stream(device).synchronize()      # RCCL has delivered before the apply kernels read `out`
        return [o[: n[r]] for r, o in enumerate(out)]
    return allgather_device


class ShardedEngine:
    def __init__(self, sc: Scenario, num_parts: int, seed: int, rank: int = 0, world: int = 1, device: int = 0, use_lds: bool = True,
                 allreduce: Optional[Callable[[np.ndarray, str], np.ndarray]] = None, trace_moves: int = 0, t_step: Optional[float] = None,
                 max_part_nodes: int = 0, allgather_bytes: Optional[Callable[[np.ndarray], List[np.ndarray]]] = None, device_tree: bool = False,
                 allgather_device: Optional[Callable] = None):
        self.sc, self.num_parts_requested, self.seed, self.rank, self.world = sc, num_parts, seed, rank, world
        self.t_step = t_step if t_step is not None else sc.default_t_step()
        if world == 1:
            allreduce = allreduce or (lambda a, op: a)
            allgather_bytes = allgather_bytes or (lambda b: [b])
        elif allreduce is None or allgather_bytes is None:
            ar, ag = _torch_collectives("cuda:%d" % device)
            allreduce, allgather_bytes = allreduce or ar, allgather_bytes or ag
            if allgather_device is None and device_tree:
                allgather_device = _torch_device_allgather("cuda:%d" % device)   # RCCL: the node exchange of a cycle never leaves the devices
        self.allreduce, self.allgather_bytes = allreduce, allgather_bytes
        self.allgather_device = allgather_device   # (nbytes, fill(tensor)) -> [device tensor per rank], or None: exchange through host buffers
        self.backend = EmatBackend(sc.num_sites, device=device, use_lds=use_lds, trace_moves=trace_moves)
        self.run = EmatRun(self.backend, sc.tree, sc.ref, seed)
        self.total_parts = 0
        self.num_local_parts = 0
        self.part_lo = self.part_hi = 0
        self.root_part = -1
        self.topology = True
        self.only_displace = False
        self.max_part_nodes = max_part_nodes   # not in the reference: cut larger parts further (0 = the reference's rule)
        self.device_tree = device_tree         # every rank keeps the whole tree in its HBM; ranks exchange node updates, not part trees
        self._configured = False
        self.spans = None                      # {phase: seconds} when a caller wants a cycle's time booked by phase (bench.py `inclusive` at N > 1)

    def _lap(self, name, t0):
        """Books the time since t0 under `name` (only when self.spans is a dict) and returns now."""
        import time
        t1 = time.perf_counter()
        if self.spans is not None:
            self.spans[name] = self.spans.get(name, 0.0) + (t1 - t0)
        return t1

    def close(self):
        self.run.close()
        self.backend.close()

    def _configure(self):
        sc, run = self.sc, self.run
        run.set_num_parts(self.num_parts_requested)
        run.set_max_part_nodes(self.max_part_nodes)
        run.set_hky(sc.mu, sc.kappa, sc.pi, sc.nu_l)
        run.set_pop_model(sc.pop)
        run.set_coalescent_t_step(self.t_step)
        run.set_flags(self.only_displace, self.topology)
        run.set_shard(self.rank, self.world)
        if self.device_tree:
            run.set_device_tree(True)
        self._configured = True

    def repartition(self):
        """Cut the tree (identically on every rank), upload this rank's parts, build their coalescent parts across ranks."""
        import time
        if not self._configured:
            self._configure()
        t = time.perf_counter()
        self.run.repartition()
        self.total_parts, self.root_part = self.run.num_parts()
        self.part_lo, self.part_hi, self.local_root = self.run.shard_range()
        self.num_local_parts = self.part_hi - self.part_lo
        t = self._lap("1 repartition (stencil, partition_tree, cut-point states, slabs of the local block)", t)
        if not self.device_tree:      # (with the tree on the devices every rank builds the whole grid itself, identically: no exchange)
            self.build_coalescent()
            self._lap("1b coalescent grid (three all-reduces)", t)

    def setup(self):
        """First cut + upload (what bench.py and the probes call before timing resident passes)."""
        self.repartition()
        self.local_sizes = [self.backend.part_stats(p)["num_nodes"] if self.device_tree else self.run.part(self.part_lo + p)[0].num_nodes for p in range(self.num_local_parts)]

    def build_coalescent(self):
        """very_scalable_coalescent.cpp:85-232 with its three cross-part reductions done as all-reduces."""
        b = self.backend
        if self.world == 1:
            return   # emat_run_repartition built them in one go
        lo, hi = self.run.coalescent_begin()
        lo = float(self.allreduce(np.array([lo]), "min")[0])
        hi = float(self.allreduce(np.array([hi]), "max")[0])
        b.coalescent_set_range(lo, hi)
        k_bar, num_active = b.coalescent_local_grid()
        k_bar = self.allreduce(k_bar, "sum")
        num_active = self.allreduce(num_active.astype(np.int64), "sum").astype(np.int32)
        k_tw = b.coalescent_sample(k_bar, num_active)
        k_tw = self.allreduce(k_tw, "sum")
        b.coalescent_finish(k_tw)

    def reassemble(self):
        """Every rank receives every other rank's parts and gathers the same whole tree (reference run.cpp:195-256)."""
        import time
        t0 = time.perf_counter()
        if self.device_tree and self.world > 1:
            # the trees stay on the devices: the rank with the root part publishes how the root sequence changed, every rank gathers
            # its own parts into its own copy of the tree, and the ranks exchange what their parts own (include/emat_backend.h)
            b = self.backend
            rd = b.tree_root_deltas()
            t0 = self._lap("3 wait for the pass + root-sequence changes", t0)
            mine = np.zeros(0, np.uint8) if rd is None else np.concatenate([np.array([len(rd[0])], np.int32).view(np.uint8), rd[0].view(np.uint8), rd[1], rd[2]])
            owner = [g for g in self.allgather_bytes(mine) if g.shape[0] > 0]
            assert len(owner) == 1, "exactly one rank holds the root part"
            k = int(owner[0][:4].view(np.int32)[0])
            site = owner[0][4:4 + 4 * k].view(np.int32).copy(); frm = owner[0][4 + 4 * k:4 + 5 * k].copy(); to = owner[0][4 + 5 * k:4 + 6 * k].copy()
            t0 = self._lap("4 all-gather of the root-sequence changes (small)", t0)
            b.tree_gather_local(site, frm, to)
            t0 = self._lap("5 gather of the local parts into the local copy of the tree", t0)
            nbytes = 0
            if self.allgather_device is not None:
                # device buffers end to end: the export kernels write into the tensor RCCL sends, the apply kernels read what it delivered
                need = b.tree_export_size()
                got = self.allgather_device(need, lambda mine: b.tree_export_nodes_into(mine.data_ptr(), int(mine.numel())))
                t0 = self._lap("6 export of the local nodes + all-gather (device buffers)", t0)
                for r, t in enumerate(got):
                    nbytes += int(t.numel())
                    if r != self.rank:
                        b.tree_apply_nodes_at(t.data_ptr(), int(t.numel()))
            else:
                exported = b.tree_export_nodes()
                got = self.allgather_bytes(exported)
                t0 = self._lap("6 export of the local nodes + all-gather (host buffers)", t0)
                for r, buf in enumerate(got):
                    nbytes += int(buf.shape[0])
                    if r != self.rank:
                        b.tree_apply_nodes(buf)
            self.last_exchange_bytes = nbytes
            t0 = self._lap("7 apply of the other ranks' nodes", t0)
            b.tree_reassemble_end()
            self.run.note_device_reassembled(site, to)
            self._lap("8 children mirror to the host, driver state", t0)
            return
        if self.world > 1:
            mine = self.run.pack_local_parts()
            t0 = self._lap("3 wait for the pass + pack of the local parts", t0)
            got = self.allgather_bytes(mine)
            self.last_exchange_bytes = int(sum(g.shape[0] for g in got))
            t0 = self._lap("6 all-gather of the serialised parts (host buffers)", t0)
            for r, buf in enumerate(got):
                if r != self.rank:
                    self.run.unpack_parts(buf)
            t0 = self._lap("7 unpack of the other ranks' parts", t0)
        self.run.reassemble()
        self._lap("8 reassemble", t0)

    def cycle(self, local_moves: int):
        """One cycle of reference Run::do_mcmc_steps without its global moves (run.cpp:622-657)."""
        import time
        self.repartition()
        t = time.perf_counter()
        self.run.run_moves_sharded(local_moves)
        if self.spans is not None:           # (booked apart from the reassemble, which would wait for the kernels anyway: same critical path)
            t = self._lap("2a launch of the pass", t)
            self.backend.synchronize()
            self._lap("2b the pass (kernels of the local block)", t)
        self.reassemble()
        t = time.perf_counter()
        tot = self.global_totals()
        self._lap("9 totals (all-reduce of two doubles)", t)
        return tot

    def tree(self):
        return self.run.tree()

    def local_stats(self):
        tot = dict(algorithmic_bytes=0, moves_done=0, bad_parts=0, proposed=[0] * 5, accepted=[0] * 5, algorithmic_bytes_of_part=[], algorithmic_write_bytes_of_part=[])
        for p in range(self.num_local_parts):
            s = self.backend.part_stats(p)
            tot["algorithmic_bytes"] += s["algorithmic_bytes"]
            tot["algorithmic_bytes_of_part"].append(s["algorithmic_bytes"]); tot["algorithmic_write_bytes_of_part"].append(s["algorithmic_write_bytes"])
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

    def Ttwiddle_l(self):
        """calc_Ttwiddle_l of the whole tree from parts spread over the ranks (reference phylo_tree_calc.cpp:176-222)."""
        b = self.backend
        lengths = np.zeros(self.total_parts)
        lengths[self.part_lo: self.part_hi] = b.part_tree_lengths(self.num_local_parts)
        lengths = self.allreduce(lengths, "sum")                                  # every rank needs every part's length
        off, node, val = self.run.Ttwiddle_ext(lengths, self.num_local_parts)
        S, R, T = b.Ttwiddle_l_partial(off, node, val)
        S = self.allreduce(S, "sum"); R = self.allreduce(R, "sum"); T = float(self.allreduce(np.array([T]), "sum")[0])
        return b.Ttwiddle_l_finish(S, R, T)

    def scalable_coalescent_log_prior(self, t_ref: float):
        """Whole-tree grid prior (reference Run::calc_cur_log_coalescent_prior) from parts spread over the ranks."""
        b = self.backend
        _, _, first = b.scalable_coalescent_partial(t_ref, self.t_step, 0, 0)
        first = int(self.allreduce(np.array([first], np.int64), "min")[0])
        kb, logs, _ = b.scalable_coalescent_partial(t_ref, self.t_step, first, -first)
        kb = self.allreduce(kb, "sum")
        logs = float(self.allreduce(np.array([logs]), "sum")[0])
        return b.scalable_coalescent_log_prior_from_grid(t_ref, self.t_step, first, kb, logs)
