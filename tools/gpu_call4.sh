#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_render.py tests/test_gpu_train_step.py tests/test_gpu_kernels.py -m gpu -q --timeout 1200 -s 2>&1 | tail -150 > $OUT/c4_tests.log
tail -12 $OUT/c4_tests.log
