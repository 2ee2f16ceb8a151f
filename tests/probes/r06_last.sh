#!/bin/bash
# last GPU call of round 6: the full suite once on the final tree (the ops wrappers gained argument checks after the soak's
# snapshot), the pre-heat A/B of bench.py, then the two plain bench lines of the final tree (hbm_bytes_per_step fix, bounded
# all-cores CPU baseline, pre-heat)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
SNR_POISON_WS=1 timeout 1500 python -m pytest tests -q -m gpu --tb=long -rA --durations=25 -p no:cacheprovider > $OUT/r06_last_suite.txt 2>&1
echo "suite exit=$? : $(tail -1 $OUT/r06_last_suite.txt)"; grep "^FAILED\|^ERROR" $OUT/r06_last_suite.txt | head
: > $OUT/r06_preheat_ab.txt
for rep in 1 2; do for ph in 0 1.5 4; do
  python bench.py --preheat-s $ph --steps 20 --warmup 5 --no-cpu-baseline --no-hashgrid --no-frame > /tmp/b.log 2>&1
  python - "$ph" >> $OUT/r06_preheat_ab.txt <<'PY'
import json, sys
for l in open("/tmp/b.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("preheat %-4s s: ms/step (median of 5 blocks of 20) %.4f  blocks %s  sustained %.4f  pair %.4f" % (sys.argv[1], d["ms_per_step"], [round(b / 20, 4) for b in d["block_ms"]], d["sustained"]["ms_per_step"], d["kernels"]["mlp_wgrad_pair"]["ms_per_step"]))
PY
done; done
cat $OUT/r06_preheat_ab.txt
python bench.py > $OUT/r06_bench_default.json 2> $OUT/r06_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_driver_style.json 2> $OUT/r06_bench_driver_style.err
tail -c 600 $OUT/r06_bench_default.json; echo; tail -c 300 $OUT/r06_bench_driver_style.json
