#!/bin/bash
# round-4 measurement batch: what the scans without a limit cost the slowest chains; what non-uniform addressing would cost; the heavy parts by function
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
for v in base no_unlimited; do
  echo "== $v" >> $O/tail.txt
  EMAT_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/$v.so timeout 300 python scripts/tail_probe.py 2>&1 | head -24 >> $O/tail.txt
done
timeout 600 bash scripts/ab_paths.sh 2 build/variants/base.so build/variants/divergent.so build/variants/no_unlimited.so > $O/ab.txt 2>&1
EMAT_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/prof.so EMAT_FN_MIN_LISTS=2400 timeout 400 python scripts/gpu_probe.py fn > $O/fn_heavy.txt 2>&1
EMAT_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/prof.so timeout 400 python scripts/gpu_probe.py fn > $O/fn_all.txt 2>&1
EMAT_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/unlim_probe.so timeout 400 python scripts/gpu_probe.py slowphase > $O/slowphase_unlim.txt 2>&1
tail -5 $O/tail.txt; cat $O/ab.txt; head -30 $O/fn_heavy.txt
