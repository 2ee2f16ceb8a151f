#!/bin/bash
# Round 4, experiment 1: can a FEW CUs pull the plain weight-gradient jobs' stream while the rest run the pair kernel?
#   (a) the plain kernel alone on 16 / 32 / 48 / 64 workgroups (SNR_WGRAD_SPLITS), (b) the pair kernel on 112 / 104 / 96 slots
# Everything in one gpurun call (box-to-box variance).  Output: gpurun_out/r04_cu_split.txt
OUT=gpurun_out/r04_cu_split.txt
mkdir -p gpurun_out
: > $OUT
B="python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-hashgrid --no-frame"
summ() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); k = d["kernels"]
        print("step %.4f  " % d["ms_per_step"] + "  ".join("%s %.4f" % (n.replace("mlp_", ""), k[n]["ms_per_step"]) for n in ("mlp_fwd", "mlp_dgrad", "mlp_wgrad_pair", "mlp_wgrad", "mlp_wgrad_reduce") if n in k))
PY
}
echo "== base" >> $OUT; $B > /tmp/b.log 2>&1; summ /tmp/b.log >> $OUT
for s in 16 32 48 64 128; do
  echo "== SNR_WGRAD_SPLITS=$s" >> $OUT; SNR_WGRAD_SPLITS=$s $B > /tmp/b.log 2>&1; summ /tmp/b.log >> $OUT
done
for s in 120 112 104 96; do
  echo "== SNR_PAIR_SLOTS=$s" >> $OUT; SNR_PAIR_SLOTS=$s $B > /tmp/b.log 2>&1; summ /tmp/b.log >> $OUT
done
echo "== SNR_RECOMPUTE=0" >> $OUT; SNR_RECOMPUTE=0 $B > /tmp/b.log 2>&1; summ /tmp/b.log >> $OUT
echo "== base again" >> $OUT; $B > /tmp/b.log 2>&1; summ /tmp/b.log >> $OUT
cat $OUT
