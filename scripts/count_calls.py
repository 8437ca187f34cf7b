#!/usr/bin/env python3
"""How often every device function is entered per move (and per topology move) at C4.
  scripts/count_calls.py build   here: copies delphy_amd/csrc to /tmp, puts EMAT_CALLED(header) at the top of every device
                                 function of the three move headers and compiles delphy_amd/libemat_calls.so (-DEMAT_COUNT_CALLS)
  scripts/count_calls.py run     on the GPU box: two passes of C4 with that library, calls per function
Why: with one lane active, every NON-LEAF device function pays a whole-wave save of the VGPR it parks its return address in
(scripts/micro/wwm.hip: about 490 cycles per call); the counts say which helpers are worth inlining or making leaves."""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["emat_device_core.hpp", "emat_device_spr.hpp", "emat_device_moves.hpp"]   # header ids as in EMAT_TIMED
SIG = re.compile(r"^(template <[^>]*> )?(EMAT_NOTAIL )?EMAT_DN? ")   # not the force-inlined accessors


def instrument(text, fid):
    lines = text.split("\n"); n = 0
    for i, l in enumerate(lines):
        if not SIG.match(l): continue
        j = i; buf = l
        while "{" not in buf and ";" not in buf and j + 1 < len(lines): j += 1; buf += "\n" + lines[j]
        k = buf.find("("); depth = 0; end = -1
        for q in range(k, len(buf)):
            if buf[q] == "(": depth += 1
            elif buf[q] == ")":
                depth -= 1
                if depth == 0: end = q; break
        if end < 0: continue
        rest = buf[end + 1:]
        if rest.lstrip().startswith(";") or "{" not in rest: continue
        b = end + 1 + rest.index("{")
        buf = buf[:b + 1] + " EMAT_CALLED(%d);" % fid + buf[b + 1:]
        new = buf.split("\n")
        # the counter key is the line of the macro: keep it on the line of the opening brace
        lines[i:j + 1] = new; n += 1
    return "\n".join(lines), n


def build():
    tmp = "/tmp/emat_cc"; shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp + "/delphy_amd"); shutil.copytree(ROOT + "/delphy_amd/csrc", tmp + "/delphy_amd/csrc"); shutil.copytree(ROOT + "/include", tmp + "/include")
    for fid, f in enumerate(FILES):
        p = tmp + "/delphy_amd/csrc/" + f
        t, n = instrument(open(p).read(), fid); open(p, "w").write(t); print(f, n, "functions instrumented")
    out = ROOT + "/delphy_amd/libemat_calls.so"
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "-Wno-unused-result", "-mllvm", "-amdgpu-lower-module-lds-strategy=module", "-DEMAT_COUNT_CALLS",
           '-DEMAT_BUILD_ID="calls"', "-shared", "-o", out, "emat_backend.hip", "emat_run.cpp", "emat_dphy.cpp", "emat_multi.cpp", "-ldl"]
    subprocess.run(cmd, cwd=tmp + "/delphy_amd/csrc", check=True, stderr=subprocess.DEVNULL)
    print("built", out)


def run(nparts=8192, moves=1000):
    import ctypes as C
    import numpy as np
    os.environ["EMAT_LIB_PATH"] = ROOT + "/delphy_amd/libemat_calls.so"
    sys.path.insert(0, ROOT)
    import delphy_amd as d
    from delphy_amd.scenarios import make_scenario
    from delphy_amd.sharding import ShardedEngine
    sc = make_scenario("C4")
    eng = ShardedEngine(sc, num_parts=nparts, seed=20261001, rank=0, world=1, device_tree=False, allreduce=lambda a, op: a, allgather_bytes=lambda b: [b])
    eng.setup()
    lib = d.load_library()
    lib.emat_debug_fn_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    buf = (C.c_uint64 * 12288)()
    eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    assert lib.emat_debug_fn_ticks(eng.backend.handle, buf) == 0   # clears
    prop0 = np.array([eng.backend.part_stats(p)["proposed"] for p in range(eng.num_local_parts)], dtype=np.float64).sum(axis=0)
    eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
    assert lib.emat_debug_fn_ticks(eng.backend.handle, buf) == 0
    prop = np.array([eng.backend.part_stats(p)["proposed"] for p in range(eng.num_local_parts)], dtype=np.float64).sum(axis=0) - prop0
    nmoves = eng.num_local_parts * moves; ntopo = prop[3] + prop[4]
    a = np.array(list(buf), dtype=np.float64).reshape(3 * 2048, 2)[:, 1]
    # names from the UNinstrumented headers: line numbers are the same (the macro goes on the brace's line)
    src = [open(os.path.join(ROOT, "delphy_amd", "csrc", f)).read().split("\n") for f in FILES]
    rows = []
    for k in np.nonzero(a)[0]:
        f, line = divmod(int(k), 2048)
        text = " ".join(src[f][max(0, line - 2):line])
        m = re.search(r"EMAT_D[NF]? .*?(\w+)\s*\(", src[f][line - 1]) or re.search(r"EMAT_D[NF]? .*?(\w+)\s*\(", text)
        kind = "DN" if "EMAT_DN" in text else ("DF" if "EMAT_DF" in text else "D")
        rows.append((a[k], (m.group(1) if m else text[:40]), kind, "%s:%d" % (FILES[f].replace("emat_device_", ""), line)))
    rows.sort(reverse=True)
    print("parts %d, %d moves (%d topology moves: %d slides, %d SPR1)" % (eng.num_local_parts, nmoves, ntopo, prop[3], prop[4]))
    print("%-44s %4s %-16s %12s %14s" % ("function", "kind", "where", "calls/move", "calls/topomove"))
    for n, name, kind, where in rows:
        print("%-44s %4s %-16s %12.3f %14.2f" % (name, kind, where, n / nmoves, n / ntopo))
    eng.close()


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
