cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-frame --no-hashgrid --blocks 3"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py tests/test_gpu_train_step.py tests/test_gpu_edge_sizes.py -q -x --timeout 600 > gpurun_out/ab_tests.txt 2>&1; tail -3 gpurun_out/ab_tests.txt
for i in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_base.so; else unset SNR_LIB; fi
    timeout 300 $B 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']
print('$v', round(d['ms_per_step'],4), 'pair', round(k['mlp_wgrad_pair']['ms_per_step'],4), 'fwd', round(k['mlp_fwd']['ms_per_step'],4), 'dgrad', round(k['mlp_dgrad']['ms_per_step'],4), 'wgrad', round(k['mlp_wgrad']['ms_per_step'],4), 'reduce', round(k['mlp_wgrad_reduce']['ms_per_step'],4))
" | tee -a gpurun_out/ab_result.txt
  done
done
