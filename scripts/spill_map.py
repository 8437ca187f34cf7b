#!/usr/bin/env python3
"""Static map of private-memory (scratch) instructions in the gfx950 ISA of the engine: per function, which source
lines the register allocator's spill/reload code and stack objects are attributed to.
Usage: scripts/spill_map.py [function-substring ...]   (compiles delphy_amd/csrc/emat_backend.hip with line tables)"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "delphy_amd", "csrc", "emat_backend.hip")
out = "/tmp/emat_g.s"
extra = os.environ.get("EMAT_EXTRA_FLAGS", "").split()
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-gline-tables-only", "-S", "-o", out, src] + extra,
               check=True, stderr=subprocess.DEVNULL)
want = sys.argv[1:]
files = {}; fn = None; loc = None
per_fn = collections.defaultdict(collections.Counter); tot = collections.Counter()
for l in open(out):
    m = re.match(r'^(_Z\w+):', l)
    if m: fn = m.group(1); loc = None; continue
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]; continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    t = l.strip()
    if fn and (t.startswith('scratch_load') or t.startswith('scratch_store')):
        per_fn[fn][loc] += 1; tot[fn] += 1
names = subprocess.run(['c++filt'], input='\n'.join(tot.keys()), capture_output=True, text=True).stdout.split('\n')
dem = dict(zip(tot.keys(), names))
for fn, n in tot.most_common():
    d = dem[fn].replace('emat::dev::', '')
    if want and not any(w in d for w in want): continue
    print("%4d  %s" % (n, d[:100]))
    if want:
        for loc, k in per_fn[fn].most_common(25): print("        %3d  %s:%s" % (k, loc[0] if loc else None, loc[1] if loc else None))
