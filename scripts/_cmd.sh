python -m pytest tests -m gpu -x -q 2>&1 | tail -2
EMAT_VERBOSE=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "classes|value" | sed -E 's/.*classes: (.*)/\1/; s/.*"value": ([0-9.]+).*"ms_per_step": ([0-9.]+).*/value \1 ms \2/'
