#!/bin/bash
# closing GPU call of round 6: the two plain bench lines of the final bench.py, then sequential poisoned full-suite passes
# (they include the new 2-rank overlapped-all-reduce test) until the budget is used
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
python bench.py > $OUT/r06_bench_default.json 2> $OUT/r06_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_driver_style.json 2> $OUT/r06_bench_driver_style.err
tail -c 420 $OUT/r06_bench_default.json; echo
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tests/probes/r06_soak_timed.sh ${SOAK_SECONDS:-2400} 1 32 closing
