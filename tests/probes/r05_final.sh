#!/bin/bash
# the round's evidence in one gpurun call: rocprofv3 PMC passes + trace of the default bench command, then the plain default
# bench line, then the driver's command line.  The PMC file is copied into profiles/ BEFORE the lines that quote it are made,
# and the trace pass is repeated behind it, so that every committed line's hbm_bytes_per_step / roofline.traffic come from
# this round's counters.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
bash tools/profile.sh r05 > $OUT/r05_profile.log 2>&1
cp $OUT/r05_pmc.json profiles/r05_pmc.json
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf $OUT/r05_trace
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05_trace -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-frame --no-hashgrid > $OUT/r05_bench_under_rocprof.json 2> $OUT/r05_trace.err
python3 tools/profile_summary.py $OUT r05 > $OUT/r05_summary.md
python bench.py > $OUT/r05_bench_default.json 2> $OUT/r05_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_driver_style.json 2> $OUT/r05_driver_style.err
tail -c 300 $OUT/r05_bench_default.json
