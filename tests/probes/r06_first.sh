#!/bin/bash
# first GPU call of round 6: one poisoned full-suite pass with full output, the determinism loops (plain and poisoned), the
# fp32 gate measurements, then more poisoned passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tests/probes/r06_soak.sh 1 1 first
timeout 1200 python tests/probes/r06_determinism.py 1.0 > gpurun_out/r06_determinism.txt 2>&1; echo "determinism exit $?"; tail -5 gpurun_out/r06_determinism.txt
SNR_POISON_WS=1 timeout 900 python tests/probes/r06_determinism.py 0.3 > gpurun_out/r06_determinism_poison.txt 2>&1; echo "determinism(poison) exit $?"; tail -3 gpurun_out/r06_determinism_poison.txt
timeout 600 python tests/probes/r06_fp32_gates.py > gpurun_out/r06_fp32_gates.jsonl 2> gpurun_out/r06_fp32_gates.err; echo "gates exit $?"
bash tests/probes/r06_soak.sh 6 1 soakA
