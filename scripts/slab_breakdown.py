#!/usr/bin/env python3
"""Byte breakdown of the staged part state at a workload (no GPU needed: host-only handle).  Usage: slab_breakdown.py [C4] [parts]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import configure, split_parts

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
sc = make_scenario(name)
parts, incl, seeds, root_part, ref = split_parts(sc, nparts, 20261001)
b = d.EmatBackend(sc.num_sites, device=-1)
configure(b, sc, ref, parts, incl, seeds, root_part)
rows = [b.debug_slab_layout(p) for p in range(len(parts))]
keys = ("header", "nodes", "cells", "trace", "heap_used", "heap_cap", "scratch", "num_cells")
A = np.array([[r[k] for k in keys] for r in rows], np.float64)
staged = A[:, 0] + A[:, 1] + A[:, 2] + A[:, 3] + A[:, 4]
print("%s: %d parts, nodes per part p50 %d p90 %d max %d" % (name, len(parts), np.median(A[:, 1] / 64), np.percentile(A[:, 1] / 64, 90), A[:, 1].max() / 64))
print("%-10s %8s %8s %8s %8s %8s" % ("bytes", "mean", "p50", "p60", "p90", "p99"))
for i, k in enumerate(keys):
    c = A[:, i]
    print("%-10s %8.0f %8.0f %8.0f %8.0f %8.0f" % (k, c.mean(), np.median(c), np.percentile(c, 60), np.percentile(c, 90), np.percentile(c, 99)))
print("%-10s %8.0f %8.0f %8.0f %8.0f %8.0f" % ("staged", staged.mean(), np.median(staged), np.percentile(staged, 60), np.percentile(staged, 90), np.percentile(staged, 99)))
b.close()
