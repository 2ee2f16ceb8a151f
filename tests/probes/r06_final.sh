#!/bin/bash
# the round's measurement set in one gpurun call (as tests/probes/r05_final.sh): PMC passes + kernel trace of the default bench
# command, the PMC file copied into profiles/ BEFORE the lines that quote it are made, then the default bench line and the
# driver's command line; beside them the pre-registered PSNR block, the byte-identity loops at 5x, the tests that changed.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
REUSE_FP32=1 SEED0=0 SEEDS=64 timeout 2400 python tests/probes/psnr_r04.py > $OUT/r06_psnr_heldout_raw.txt 2>&1; echo "psnr exit $?"; tail -8 $OUT/r06_psnr_heldout_raw.txt
SNR_POISON_WS=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nccl_one_rank.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -3
timeout 1200 python tests/probes/r06_determinism.py 5.0 > $OUT/r06_determinism_x5.txt 2>&1; echo "determinism x5 exit $?"; tail -2 $OUT/r06_determinism_x5.txt
bash tools/profile.sh r06 > $OUT/r06_profile.log 2>&1
cp $OUT/r06_pmc.json profiles/r06_pmc.json
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf $OUT/r06_trace
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_trace -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-frame --no-hashgrid > $OUT/r06_bench_under_rocprof.json 2> $OUT/r06_trace.err
python3 tools/profile_summary.py $OUT r06 > $OUT/r06_summary.md
python bench.py > $OUT/r06_bench_default.json 2> $OUT/r06_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_driver_style.json 2> $OUT/r06_bench_driver_style.err
tail -c 400 $OUT/r06_bench_default.json; echo; tail -c 300 $OUT/r06_bench_driver_style.json
find $OUT/r06_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r06_kernel_stats.csv
find $OUT/r06_hg_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r06_hg_kernel_stats.csv
