#!/bin/bash
# Sanitizer runs of the HOST side (CPU build only; GPU sanitizers are not available on the pool): the C++ run driver
# (partitioning, refinement, subtree building, thread pool, reassemble) is compiled with g++ -fsanitize against stubs of
# the backend entry points and driven through several repartition / part_put / reassemble cycles.
# Usage: scripts/sanitize_host.sh        (runs ASan+UBSan, then TSan)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
python3 - "$ROOT" "$W" <<'PY'
import re, sys
root, w = sys.argv[1], sys.argv[2]
# every entry point the three headers declare gets a WEAK stub: what emat_run.cpp defines wins, the rest (engine, emat_multi, .dphy writer)
# answers "no device" -- the Python mirror resolves every symbol when it loads a library
out = ['#include "%s/include/emat_backend.h"' % root, '#include "%s/include/emat_host.h"' % root, '#include "%s/include/emat_dphy.h"' % root, 'extern "C" {']
seen = set()
for h in ("emat_backend.h", "emat_host.h", "emat_dphy.h"):
    hdr = re.sub(r'/\*.*?\*/', '', open(root + "/include/" + h).read(), flags=re.S)
    for ret, name, args in re.findall(r'^((?:const\s+)?[A-Za-z_][\w]*\s*\*?)\s+(emat_\w+)\s*\(([^;{]*)\)\s*;', hdr, flags=re.M | re.S):
        if name in seen: continue
        seen.add(name)
        r = ret.strip()
        body = "return EMAT_ERR_NO_DEVICE;" if r == "emat_status" else ('return "stub";' if r.replace(" ", "") == "constchar*" else ("" if r == "void" else ("return nullptr;" if r.endswith("*") else "return 0;")))
        out.append('__attribute__((weak)) %s %s(%s) { %s }' % (r, name, args.strip(), body))
out += ['}']
open(w + "/stubs.cpp", "w").write("\n".join(out))
open(w + "/drive.py", "w").write('''
import sys; sys.path.insert(0, "%s"); sys.path.insert(0, "%s/tests")
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import assert_trees_match
sc = make_scenario("C2", num_tips=3000, num_sites=5000, uncertain_tips=0.2)
for limit in (0, 40):
    run = d.EmatRun(None, sc.tree, sc.ref, 3); run.set_num_parts(200); run.set_max_part_nodes(limit)
    for cyc in range(5):
        run.repartition(); n, rp = run.num_parts()
        parts = [run.part(i) for i in range(n)]
        for i in range(0, n, 7): run.part_put(i, parts[i][0])
        run.reassemble()
    t2, ref2 = run.tree(); assert_trees_match(t2, sc.tree, 0.0, "sanitizer drive")
    run.close()
print("host driver: OK")
''' % (root, root))
PY
for san in "address,undefined" "thread"; do
  g++ -O1 -g -std=c++17 -fsanitize=$san -fno-omit-frame-pointer -fPIC -shared -pthread -o $W/libemat_san.so $ROOT/delphy_amd/csrc/emat_run.cpp $W/stubs.cpp
  if [ "$san" = "thread" ]; then PRE=$(g++ -print-file-name=libtsan.so); else PRE=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so); fi
  echo "== -fsanitize=$san"
  LD_PRELOAD=$PRE ASAN_OPTIONS=detect_leaks=0 TSAN_OPTIONS="report_signal_unsafe=0" EMAT_HOST_THREADS=8 EMAT_LIB_PATH=$W/libemat_san.so python3 $W/drive.py 2>&1 | grep -E "ERROR|WARNING: ThreadSanitizer|runtime error|host driver" | sort | uniq -c
done
# the whole library (C-ABI, slab encode / decode, coalescent-part builder) with HOST-side ASan, through the CPU tests that
# use host-only handles (the default initial-tree builder among them: host code throughout); the device code is compiled without instrumentation
echo "== hipcc -fsanitize=address -fno-gpu-sanitize (host code of libemat_hip.so), CPU tests"
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -Wno-unused-function -Wno-unused-result \
      -shared -o $W/libemat_hip_asan.so $ROOT/delphy_amd/csrc/emat_backend.hip $ROOT/delphy_amd/csrc/emat_run.cpp $ROOT/delphy_amd/csrc/emat_dphy.cpp $ROOT/delphy_amd/csrc/emat_multi.cpp -ldl
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
(cd $ROOT && LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 EMAT_LIB_PATH=$W/libemat_hip_asan.so python3 -m pytest tests/test_host_driver.py tests/test_abi.py tests/test_dphy_writer.py tests/test_list_limits.py tests/test_initial_tree.py -x -q -m "not gpu" 2>&1 | grep -E "AddressSanitizer|passed|failed")
rm -rf $W
