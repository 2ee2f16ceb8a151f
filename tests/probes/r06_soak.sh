#!/bin/bash
# Round 6 soak (VERDICT r05 "next round" item 1).  Every pass keeps its FULL pytest output (--tb=long -rA) in its own file,
# so that a failing assertion is on record whether or not it reproduces.
#   r06_soak.sh <passes> [poison=1] [tag]
# poison=1 runs the suite with SNR_POISON_WS=1 (spin-nerf_amd/_debug.py: every torch.empty buffer 0xFF-filled first).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-5}; P=${2:-1}; TAG=${3:-soak}
OUT=gpurun_out/r06_$TAG; mkdir -p $OUT
for i in $(seq 1 $N); do
  SNR_POISON_WS=$P timeout 1500 python -m pytest tests -q -m gpu --tb=long -rA -p no:cacheprovider > $OUT/pass_$i.txt 2>&1
  echo "pass $i poison=$P exit=$? : $(tail -1 $OUT/pass_$i.txt)"
  if grep -q "^FAILED\|^ERROR" $OUT/pass_$i.txt; then grep "^FAILED\|^ERROR" $OUT/pass_$i.txt; else gzip -f $OUT/pass_$i.txt; fi
done
