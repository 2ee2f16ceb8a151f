#!/bin/bash
# Collect the rocprofv3 evidence bench.py's numbers are judged against (run on the GPU box):
#   tools/profile.sh <tag>        -> gpurun_out/<tag>_*   (copy the summaries into profiles/)
# Pass 1: kernel trace + stats of the default bench command.  Passes 2..4: PMC counters, each in its
# own run with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-r05}
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
BENCH="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-frame --no-hashgrid"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o bench -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_trace.err
timeout -k 5 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_INSTS_MFMA \
  --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq -o pmc -- $BENCH > /dev/null 2> $OUT/${TAG}_pmc_sq.err
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o pmc -- $BENCH > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o pmc -- $BENCH > /dev/null 2> $OUT/${TAG}_pmc_write.err
python3 tools/profile_summary.py $OUT $TAG > $OUT/${TAG}_summary.md
cat $OUT/${TAG}_summary.md
# BASELINE config 5 (hash-grid networks): the same three kinds of passes over a training loop of that path alone
HG="python3 tests/probes/hashgrid_prof.py"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_hg_trace -o hg -- $HG > $OUT/${TAG}_hg_run.txt 2> $OUT/${TAG}_hg_trace.err
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_hg_pmc_fetch -o pmc -- $HG > /dev/null 2> $OUT/${TAG}_hg_pmc_fetch.err
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_hg_pmc_write -o pmc -- $HG > /dev/null 2> $OUT/${TAG}_hg_pmc_write.err
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/${TAG}_hg_pmc_sq -o pmc -- $HG > /dev/null 2> $OUT/${TAG}_hg_pmc_sq.err
python3 tools/profile_summary.py $OUT ${TAG}_hg "$HG (1024 rays x (64+128) samples, NeRF_TCNN coarse + fine)" nojson > $OUT/${TAG}_hg_summary.md
cat $OUT/${TAG}_hg_summary.md
