python -m pytest tests -m gpu -x -q 2>&1 | tail -2
EMAT_VERBOSE=1 python scripts/gpu_probe.py cycle 100 2>&1 | grep -E "^cycle [123]|emat_run\] repart" | cut -c1-260 | tail -5
