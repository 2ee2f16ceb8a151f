#!/bin/bash
# A/B: a pass's inputs fetched through LDS one pass ahead + persistent grid (shipped) vs loads at the top of the pass
# (-DSNR_IN_PREFETCH=0), and vs the generic encoding on top of that (-DSNR_ENC_STATIC=0 = the round-4 forward), one gpurun call
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py tests/test_gpu_train_step.py tests/test_gpu_chain2.py -m gpu -x -q 2>&1 | tail -2
AB_ARGS="--steps 30 --warmup 5 --blocks 3" bash tools/ab.sh gpurun_out/r05_prefetch_ab base nopf encold
