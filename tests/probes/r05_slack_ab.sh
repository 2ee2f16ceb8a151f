#!/bin/bash
# A/B: weight ring with LDS progress words and one block of slack (-DSNR_RING_SLACK=1) vs the s_barrier per block entry, one gpurun call
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=${1:-slack}
SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$V.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -2
AB_ARGS="--steps 30 --warmup 5 --blocks 3" timeout 900 bash tools/ab.sh gpurun_out/r05_slack_ab base $V
