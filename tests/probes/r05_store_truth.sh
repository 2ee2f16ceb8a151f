#!/bin/bash
# training forward with and without its saved-tensor stores: time AND effective clock (is the stores' cost cycles or clock?)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TRAIN=1
for v in base nostore; do
  if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  timeout 200 python tests/probes/r05_dma_truth.py 2>&1 | grep -v amdgpu
  rm -rf gpurun_out/r05_st_pmc_$v
  timeout -k 5 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r05_st_pmc_$v/c -o pmc -- python3 tests/probes/r05_dma_truth.py > /dev/null 2>&1
  python tests/probes/r05_clock_from_pmc.py gpurun_out/r05_st_pmc_$v
done
