python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo
python scripts/gpu_probe.py timeline 2>&1 | head -3
