EMAT_LIB_PATH=$PWD/delphy_amd/libemat_hip_prof.so python scripts/gpu_probe.py phase 2>&1 | grep -E "^core|unused|ALL|proposed"
