for t in 1 4 16 64; do echo "== host threads $t"; EMAT_HOST_THREADS=$t EMAT_VERBOSE=1 python scripts/gpu_probe.py cycle 2>&1 | grep -E "^cycle [12]|emat_run\] repart" | tail -3; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
