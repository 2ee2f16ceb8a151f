#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
python -m pytest tests/test_gpu_render.py tests/test_gpu_train_step.py tests/test_gpu_kernels.py -m gpu -q --timeout 1200 2>&1 | tail -40 > $OUT/c2_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/c2_bench.json 2> $OUT/c2_bench.err
tail -8 $OUT/c2_tests.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/c2_bench.json'))
print(d['value'], d['ms_per_step'], d['ms_per_frame_378x504'])
print({k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})
PY
