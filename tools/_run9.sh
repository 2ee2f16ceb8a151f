cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3h
timeout 900 python -m pytest tests/test_gpu_bench_dist.py tests/test_gpu_dist_step.py -m gpu -q --timeout 600 2>&1 | tail -4
timeout 600 python bench.py > gpurun_out/r3h/bench_default.json 2> gpurun_out/r3h/bench_default.err
tail -2 gpurun_out/r3h/bench_default.err; python - <<'PY'
import json
for l in open('gpurun_out/r3h/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print({k:d[k] for k in ('value','ms_per_step','block_ms','ms_per_frame_378x504','hbm_bytes_per_step')})
        print(d['roofline'])
        print({k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})
        print(d.get('also_measured',{}).get('fp32_mode'))
        print(d.get('also_measured',{}).get('hashgrid_config5',{}).get('roofline'))
        print(d.get('cpu_baseline'))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
