cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3c
export SNR_PAIR_PAIR=1
for kd in 0 1; do
export SNR_PAIR_KIND=$kd
for v in base pa207 pa223 pa239; do
  if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  timeout 200 python bench.py --no-cpu-baseline --no-hashgrid --no-frame --steps 20 --warmup 5 > gpurun_out/r3c/$v.txt 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r3c/$v.txt'):
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']
        print('kind $kd', '$v', 'pair', round(k['mlp_wgrad_pair']['ms_per_step'],4))
PY
done
done
