cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hashgrid --no-frame --blocks 3"
VARS="${VARS:-new wv}"
for v in $VARS; do
  if [ $v = new ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  echo "== $v: $(timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py -q -x 2>&1 | tail -1)" | tee -a gpurun_out/ab_result.txt
done
for i in 1 2 3; do
  for v in $VARS; do
    if [ $v = new ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
    timeout 300 $B 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']
print('$v', round(d['ms_per_step'],4), 'pair', round(k['mlp_wgrad_pair']['ms_per_step'],4), 'fwd', round(k['mlp_fwd']['ms_per_step'],4), 'dgrad', round(k['mlp_dgrad']['ms_per_step'],4), 'wgrad', round(k['mlp_wgrad']['ms_per_step'],4), 'reduce', round(k['mlp_wgrad_reduce']['ms_per_step'],4))
" | tee -a gpurun_out/ab_result.txt
  done
done
