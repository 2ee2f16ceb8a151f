#!/bin/bash
# timing of chain2 ablation builds (SNR_LIB), inference launch of 196 608 samples, shipped kernel beside it in every process
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  echo "== $v"; timeout 120 python tests/probes/r05_chain2_check.py --quick $C2ARGS 2>&1 | grep -E "SNR_CHAIN2="
done; done
