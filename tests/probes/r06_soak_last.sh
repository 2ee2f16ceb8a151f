#!/bin/bash
# the round's last GPU call: the two plain bench lines of the FINAL tree, then sequential poisoned full-suite passes (full output
# kept) until the time budget is used up
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
python bench.py > $OUT/r06_bench_default.json 2> $OUT/r06_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_driver_style.json 2> $OUT/r06_bench_driver_style.err
tail -c 300 $OUT/r06_bench_default.json; echo
bash tests/probes/r06_soak_timed.sh ${SOAK_SECONDS:-3600} 1 32 seqsoak
