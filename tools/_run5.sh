cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3e; rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-frame --no-hashgrid"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS --kernel-trace -d $OUT/a -o pmc -- $B > /dev/null 2> $OUT/a.err
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/b -o pmc -- $B > /dev/null 2> $OUT/b.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/d -o pmc -- $B > /dev/null 2> $OUT/d.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/e -o pmc -- $B > /dev/null 2> $OUT/e.err
for p in a b d e; do echo "== $p"; python3 tests/probes/pmc_query.py $(find $OUT/$p -name '*.db' | head -1) mlp_ | grep "pair\|mlp_wgrad_kernel\|fwd\|dgrad"; done
