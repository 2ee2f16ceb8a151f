cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in base; do
  export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so
  echo "== $v"; timeout 200 python -m pytest tests/test_gpu_kernels.py -q -x -k "mlp_backward_bf16" 2>&1 | tail -2
done
unset SNR_LIB
echo "== new default"; timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "mlp_backward_bf16" 2>&1 | tail -2
