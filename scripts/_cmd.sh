EMAT_VERBOSE=1 python -m pytest tests -m gpu -x -q -s -k "regrown" 2>&1 | grep -E "ran out|passed|failed|Error|assert" | head
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo
