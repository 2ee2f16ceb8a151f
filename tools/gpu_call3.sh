#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
python -m pytest tests/test_gpu_render.py tests/test_gpu_train_step.py -m gpu -q --timeout 1200 2>&1 | tail -150 > $OUT/c3_tests.log
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-frame > $OUT/c3_bench_rocprof.json 2> $OUT/c3_trace.err
tail -5 $OUT/c3_tests.log
f=$(find $OUT/c3_trace -name "*kernel_stats.csv" | head -1); cut -c1-90 $f | head -14; awk -F, 'NR>1{print $2","$3","$4}' $f | head -14
