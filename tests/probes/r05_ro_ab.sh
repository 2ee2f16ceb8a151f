cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_kernels.py tests/test_gpu_edge_sizes.py tests/test_gpu_spin_iter.py tests/test_gpu_train_step.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for v in base oldsort; do
  if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  python bench.py --no-cpu-baseline --no-frame --no-hashgrid --steps 30 --warmup 5 --blocks 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']
        print('$v', 'step %.4f' % d['ms_per_step'], ' '.join('%s %.4f' % (n, k[n]['ms_per_step']) for n in ('composite_train_reg','composite_train_sample','pack_rays_sample','adam_pack','mlp_wgrad_reduce','wgrad_post')))
"
done; done
