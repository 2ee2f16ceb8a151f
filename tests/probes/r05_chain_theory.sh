#!/bin/bash
# Round 5, experiment 1: is the cost of the saved-tensor stores in the chain kernels the write-acknowledge latency that the
# counted DMA waits expose (stores and LDS-DMA loads retire through ONE in-order vmcnt)?  Variants of the shipped kernels:
#   base     as shipped
#   t16      the block-entry wait tolerates 24 more outstanding operations (racy: results garbage, timing valid)
#   nostore  no activation / d z stores          nodma   no weight DMA
#   timing   s_memtime around the block waits (per-acquire wait / barrier cycles)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for v in base t16 nostore nodma timing; do
  if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  [ "$v" = timing ] && [ $rep = 2 ] && continue
  timeout 300 python tests/probes/fwd_ablate.py 2>&1 | grep -v Warning
  timeout 300 python tests/probes/bwd_ablate.py 2>&1 | grep -v Warning
done; done
