cd $GRAFT_REPO_ROOT
for r in 1 2; do
for lib in delphy_amd/libemat_hip.so build/variants/old_refine.so; do
  EMAT_VERBOSE=spans EMAT_LIB_PATH=$GRAFT_REPO_ROOT/$lib EMAT_ALLOW_STALE_LIB=1 python bench.py --no-cpu-baseline --no-decompositions --secondary '' --steps 5 > gpurun_out/ab_incl.json 2> gpurun_out/ab_incl.err
  python3 - "$lib" "$r" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_incl.json").readline())
i = d["inclusive"]
print(sys.argv[1], "rep", sys.argv[2], "| inclusive", round(i["value"]/1e6,1), "M, ms/cycle p50", i.get("ms_per_cycle_p50"), "| resident", round(d["value"]/1e6,1))
for l in open("gpurun_out/ab_incl.err"):
    if "refine_stencil" in l or "cycle: 1 repartition" in l: print("   ", l.strip())
PY
done
done
