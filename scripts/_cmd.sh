python bench.py 2>&1 | tail -1 > gpurun_out/bench_r01.json; cut -c1-200 gpurun_out/bench_r01.json
