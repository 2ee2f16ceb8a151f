#!/bin/bash
# A/B: compile-time positional encoding (shipped) vs the generic one (-DSNR_ENC_STATIC=0), one gpurun call
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -2
AB_ARGS="--steps 30 --warmup 5 --blocks 3" bash tools/ab.sh gpurun_out/r05_enc_ab base encold
