cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3d
for w in 40 55 66 80 100 130; do
  SNR_PAIR_W0=$w timeout 200 python bench.py --no-cpu-baseline --no-hashgrid --no-frame --steps 20 --warmup 5 > gpurun_out/r3d/w$w.txt 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r3d/w$w.txt'):
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']
        print('W0=$w', 'step', round(d['ms_per_step'],4), 'pair', round(k['mlp_wgrad_pair']['ms_per_step'],4))
PY
done
