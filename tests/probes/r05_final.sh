#!/bin/bash
# the round's evidence in one gpurun call: rocprofv3 trace + PMC of the default bench command, then the plain default bench line,
# then the driver's command line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/profile.sh r05 > gpurun_out/r05_profile.log 2>&1
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_driver_style.json 2> gpurun_out/r05_driver_style.err
tail -c 400 gpurun_out/r05_bench_default.json
