python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python scripts/gpu_probe.py cycle 2>&1 | grep -E "^cycle" | cut -c1-330
