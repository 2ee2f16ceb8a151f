#!/bin/bash
# Soak of the round-5 code on one box: the full GPU suite twice, then the tests of the new paths (chain2's LDS progress-word
# protocol, the direct SPIn-NeRF iteration, the merged backward, the one-rank NCCL path) ten more times.  A protocol race
# fails intermittently when it fails: count the runs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -1; done
for i in $(seq 1 10); do timeout 900 python -m pytest tests/test_gpu_chain2.py tests/test_gpu_spin_iter.py tests/test_gpu_train_step.py tests/test_gpu_fullsize.py tests/test_gpu_nccl_one_rank.py -q -m gpu 2>&1 | tail -1; done
