#!/bin/bash
# Round 4 soak: hand-counted waits fail intermittently when they fail.  Full GPU suite twice, then the bf16 backward / train-step /
# full-size tests (the merged weight-gradient launch: pair slots + 4-wave plain workgroups, the Adam + pack kernel, the fused
# per-ray kernels) ten more times, on one box.
mkdir -p gpurun_out; OUT=gpurun_out/r04_soak.txt; : > $OUT
for i in 1 2; do python -m pytest tests -q -m gpu 2>&1 | tail -1 >> $OUT; done
for i in 1 2 3 4 5 6 7 8 9 10; do
  python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py tests/test_gpu_fullsize.py -q -m gpu -k "bf16 or backward or adam or fused or merged or step" 2>&1 | tail -1 >> $OUT
done
cat $OUT
