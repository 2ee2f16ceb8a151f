#!/bin/bash
# issue / wait counters of the fused forward and dgrad kernels:  chain_pmc.sh [lib-tag ...]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/chain_pmc; mkdir -p $OUT
for v in "$@"; do
  if [ $v = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  timeout -k 5 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC \
    --kernel-trace -d $OUT/${v}_a -o pmc -- python3 tests/probes/bwd_ablate.py > /dev/null 2> $OUT/${v}_a.err
  timeout -k 5 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS \
    --kernel-trace -d $OUT/${v}_b -o pmc -- python3 tests/probes/bwd_ablate.py > /dev/null 2> $OUT/${v}_b.err
  for p in a b; do echo "== $v $p"; python3 tests/probes/pmc_query.py $(find $OUT/${v}_$p -name '*.db' | head -1) "mlp_%grad_k"; python3 tests/probes/pmc_query.py $(find $OUT/${v}_$p -name '*.db' | head -1) "mlp_fwd"; done
done
