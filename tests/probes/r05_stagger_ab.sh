#!/bin/bash
# A/B: dgrad with the second wave of every SIMD finishing tile nt - 1 in the middle of tile nt (-DSNR_STAGGER=1), one gpurun call
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=${1:-stag}
SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$V.so timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "backward or dgrad or gradient" 2>&1 | tail -2
AB_ARGS="--steps 30 --warmup 5 --blocks 3" bash tools/ab.sh gpurun_out/r05_stagger_ab base $V
