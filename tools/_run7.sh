cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/_run6.sh
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x -k "kernels" 2>&1 | tail -2
for rep in 1 2; do for v in base nopf; do
  if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  timeout 200 python bench.py --no-cpu-baseline --no-hashgrid --no-frame > gpurun_out/r3f/$v.txt 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r3f/$v.txt'):
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']
        print('$v', 'step', round(d['ms_per_step'],4), {a:round(b['ms_per_step'],4) for a,b in k.items() if a.startswith('mlp')})
PY
done; done
