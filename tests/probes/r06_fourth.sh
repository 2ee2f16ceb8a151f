#!/bin/bash
# fourth GPU call of round 6: the test whose gate changed (full output), then the parallel soak of the final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
SNR_POISON_WS=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -rA -p no:cacheprovider -k "backward_bf16" > $OUT/r06_fourth_pre.txt 2>&1
echo "pre-check exit=$? : $(tail -1 $OUT/r06_fourth_pre.txt)"; grep "AssertionError:\|^FAILED" $OUT/r06_fourth_pre.txt | head
if grep -q "^FAILED" $OUT/r06_fourth_pre.txt; then echo "pre-check failed: no soak"; exit 1; fi
bash tests/probes/r06_soak_parallel.sh ${SOAK_STREAMS:-4} ${SOAK_PASSES:-12} 1 psoak
