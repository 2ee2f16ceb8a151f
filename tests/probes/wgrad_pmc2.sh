#!/bin/bash
# memory-path counters of the wgrad kernel: full kernel vs stream-only (w1)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/wgrad_pmc2; mkdir -p $OUT
for v in base w1; do
  if [ $v = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES \
    --kernel-trace -d $OUT/${v}_c -o pmc -- python3 tests/probes/bwd_ablate.py > /dev/null 2> $OUT/${v}_c.err
  rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum \
    --kernel-trace -d $OUT/${v}_d -o pmc -- python3 tests/probes/bwd_ablate.py > /dev/null 2> $OUT/${v}_d.err
  rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum \
    --kernel-trace -d $OUT/${v}_e -o pmc -- python3 tests/probes/bwd_ablate.py > /dev/null 2> $OUT/${v}_e.err
  for p in c d e; do echo "== $v $p"; python3 tests/probes/pmc_query.py $(find $OUT/${v}_$p -name '*.db' | head -1) mlp_wgrad_k; grep -iE "error|fail|invalid" $OUT/${v}_$p.err | head -3 | cut -c1-200; done
done
