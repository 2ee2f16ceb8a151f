cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x -k "kernels" > gpurun_out/r3b/test_kernels.txt 2>&1
tail -5 gpurun_out/r3b/test_kernels.txt
for rc in 1 0 1; do
SNR_RECOMPUTE=$rc timeout 300 python bench.py --no-cpu-baseline --no-hashgrid --no-frame > gpurun_out/r3b/bench_rc$rc.txt 2>&1
python - <<PY
import json
for l in open('gpurun_out/r3b/bench_rc$rc.txt'):
    if l.startswith('{'):
        d=json.loads(l)
        print('recompute=$rc step', round(d['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})
PY
done
