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
hdr = open(root + "/include/emat_backend.h").read()
decls = re.findall(r'^(emat_status\s+emat_\w+\s*\([^;]*\))\s*;', hdr, flags=re.M | re.S)
out = ['#include "%s/include/emat_backend.h"' % root, 'extern "C" {']
out += [re.sub(r'/\*.*?\*/', '', d, flags=re.S).strip() + ' { return EMAT_ERR_NO_DEVICE; }' for d in decls]
out += ['const char* emat_last_error(const emat_backend*) { return "stub"; }', '}']
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
