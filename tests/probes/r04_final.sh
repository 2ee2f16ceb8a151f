#!/bin/bash
# Round 4 final evidence in one gpurun call: full GPU suite, default bench line, rocprofv3 passes, recompute A/B
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r04_gpu_suite.txt
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
bash tools/profile.sh r04 > gpurun_out/r04_profile_stdout.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-hashgrid --no-frame"
: > gpurun_out/r04_recompute_ab_raw.txt
for rep in 1 2; do
  for rc in 1 0; do
    echo "== SNR_RECOMPUTE=$rc" >> gpurun_out/r04_recompute_ab_raw.txt
    SNR_RECOMPUTE=$rc $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l);k=d['kernels']
        print('step %.4f  '%d['ms_per_step']+'  '.join('%s %.4f'%(n.replace('mlp_',''),k[n]['ms_per_step']) for n in ('mlp_fwd','mlp_dgrad','mlp_wgrad_pair','mlp_wgrad','mlp_wgrad_reduce','adam','mlp_pack') if n in k))" >> gpurun_out/r04_recompute_ab_raw.txt
  done
done
cat gpurun_out/r04_gpu_suite.txt; cat gpurun_out/r04_recompute_ab_raw.txt; head -c 600 gpurun_out/r04_bench_default.json
