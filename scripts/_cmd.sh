python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py 2>&1 | tail -1 > gpurun_out/bench_r01.json; cut -c1-330 gpurun_out/bench_r01.json
bash scripts/profile.sh r01 > gpurun_out/profile_r01.log 2>&1; tail -4 gpurun_out/profile_r01.log
bash scripts/pmc_mix.sh full
python scripts/gpu_probe.py cycle 2>&1 | grep -E "^cycle" | cut -c1-330
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --parts 16384 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|[0-9]* partition parts'
