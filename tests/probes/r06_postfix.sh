#!/bin/bash
# after the fix of the intermittent test: the file alone (full output), then sequential poisoned full-suite passes for <seconds>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
SNR_POISON_WS=1 timeout 900 python -m pytest tests/test_gpu_spin_iter.py -q -m gpu --tb=long -rA --durations=8 -p no:cacheprovider > $OUT/r06_postfix_file.txt 2>&1
echo "file exit=$? : $(tail -1 $OUT/r06_postfix_file.txt)"; grep "^FAILED\|^ERROR\|AssertionError" $OUT/r06_postfix_file.txt | head; grep "s call" $OUT/r06_postfix_file.txt | head -4
if grep -q "^FAILED\|^ERROR" $OUT/r06_postfix_file.txt; then exit 1; fi
bash tests/probes/r06_soak_timed.sh ${SOAK_SECONDS:-1500} 1 32 postfix
