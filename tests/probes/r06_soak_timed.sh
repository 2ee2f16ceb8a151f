#!/bin/bash
# Round 6 soak, time-budgeted: K concurrent streams of full-suite passes (SNR_POISON_WS=1, full output kept) until <seconds> are
# used up; a pass that would start after the budget does not start.  Every stream caps its CPU threads (the 4-stream run of
# tests/probes/r06_soak_parallel.sh with 256 threads per process took 15-19 MINUTES per pass: the oracle's thread pools thrashed).
#   r06_soak_timed.sh <seconds> [streams=2] [threads=32] [tag]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
BUDGET=${1:-1800}; K=${2:-2}; TH=${3:-32}; TAG=${4:-tsoak}
OUT=gpurun_out/r06_$TAG; mkdir -p $OUT
T0=$(date +%s)
stream() {
  s=$1; i=0
  while true; do
    now=$(date +%s); if [ $((now - T0)) -ge $BUDGET ]; then break; fi
    i=$((i + 1)); f=$OUT/stream${s}_pass$i.txt
    SNR_TEST_THREADS=$TH OMP_NUM_THREADS=$TH SNR_POISON_WS=1 timeout 1500 python -m pytest tests -q -m gpu --tb=long -rA -p no:cacheprovider --basetemp=/tmp/pt_${TAG}_$s > $f 2>&1
    echo "stream $s pass $i exit=$? : $(tail -1 $f)"
    if grep -q "^FAILED\|^ERROR" $f; then grep "^FAILED\|^ERROR" $f; else gzip -f $f; fi
  done
}
for s in $(seq 1 $K); do stream $s & done
wait
echo "passes: $(ls $OUT | wc -l)  failing: $(ls $OUT | grep -c 'txt$')"
