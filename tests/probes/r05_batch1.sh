#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tests/probes/hbm_bw.py 2>&1 | grep -v amdgpu
timeout 300 python tests/probes/r05_chain2_bwd.py 2>&1 | grep -v amdgpu | tail -9
echo "== storers drop their chunks (timing only)"
SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_c2nost.so timeout 300 python tests/probes/r05_chain2_bwd.py --quick 2>&1 | grep "SNR_CHAIN2="
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "backward or bwd or full" 2>&1 | tail -5
