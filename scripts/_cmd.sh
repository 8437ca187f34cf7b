python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; done
python scripts/gpu_probe.py tail 2>&1 | grep -E "kernel|sum of"
