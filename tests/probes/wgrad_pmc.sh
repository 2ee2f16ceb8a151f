#!/bin/bash
# LDS / issue counters of the wgrad kernel: full kernel vs stream-only (w1) vs compute-only (w2)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/wgrad_pmc; mkdir -p $OUT
for v in base w1 w2; do
  if [ $v = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES \
    --kernel-trace -d $OUT/${v}_a -o pmc -- python3 tests/probes/bwd_ablate.py > /dev/null 2> $OUT/${v}_a.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC \
    --kernel-trace -d $OUT/${v}_b -o pmc -- python3 tests/probes/bwd_ablate.py > /dev/null 2> $OUT/${v}_b.err
  for p in a b; do echo "== $v $p"; python3 tests/probes/pmc_query.py $(find $OUT/${v}_$p -name '*.db' | head -1) mlp_wgrad_k; tail -3 $OUT/${v}_$p.err | cut -c1-200; done
done
