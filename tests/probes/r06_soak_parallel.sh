#!/bin/bash
# Round 6 soak, parallel form: K concurrent streams of full-suite passes on the one GPU of the box (the suite is bound by the CPU
# oracle's work, not by the GPU: K streams finish K times the passes in about the same time — and the kernels of different
# processes interleave on the device, which is the harder condition for a hand-counted wait).  Every pass keeps its full pytest
# output (--tb=long -rA); passing passes are gzipped, failing ones kept verbatim.
#   r06_soak_parallel.sh <streams> <passes per stream> [poison=1] [tag]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
K=${1:-4}; N=${2:-5}; P=${3:-1}; TAG=${4:-psoak}
OUT=gpurun_out/r06_$TAG; mkdir -p $OUT
stream() {
  s=$1
  for i in $(seq 1 $N); do
    f=$OUT/stream${s}_pass$i.txt
    SNR_POISON_WS=$P timeout 2400 python -m pytest tests -q -m gpu --tb=long -rA -p no:cacheprovider --basetemp=/tmp/pt_${TAG}_$s > $f 2>&1
    echo "stream $s pass $i poison=$P exit=$? : $(tail -1 $f)"
    if grep -q "^FAILED\|^ERROR" $f; then grep "^FAILED\|^ERROR" $f; else gzip -f $f; fi
  done
}
for s in $(seq 1 $K); do stream $s & done
wait
echo "passes: $(ls $OUT | wc -l)  failing: $(ls $OUT | grep -c 'txt$')"
