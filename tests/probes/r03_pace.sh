cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-frame --no-hashgrid --blocks 3"
echo "tests: $(timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py -q -x 2>&1 | tail -1)" | tee -a gpurun_out/pace_result.txt
for cfg in "0 2" "16 2" "8 2" "16 4" "32 3"; do
  set -- $cfg
  export SNR_PAIR_POLL=$1 SNR_PAIR_LEAD=$2
  timeout 300 $B 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']
print('poll $1 lead $2: step', round(d['ms_per_step'],4), 'pair', round(k['mlp_wgrad_pair']['ms_per_step'],4))
" | tee -a gpurun_out/pace_result.txt
  rm -rf gpurun_out/pace_pmc
  timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pace_pmc -o pmc -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-frame --no-hashgrid > /dev/null 2> gpurun_out/pace_pmc.err
  python3 - <<PY | tee -a gpurun_out/pace_result.txt
import csv,glob
f=glob.glob('gpurun_out/pace_pmc/**/*counter_collection.csv', recursive=True)
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f[0])) if 'pair' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE'] if f else []
print('   pair FETCH_SIZE avg KiB per launch:', round(sum(v)/max(1,len(v))), 'x2 =', round(2*sum(v)/max(1,len(v))*1024/1e9,3), 'GB', len(v), 'launches')
PY
done
