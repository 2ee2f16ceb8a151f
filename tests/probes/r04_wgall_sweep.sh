#!/bin/bash
# Round 4: shape of the merged weight-gradient launch — pair slots vs plain workgroups (SNR_PAIR_SLOTS / SNR_PLAIN_WGS)
OUT=gpurun_out/r04_wgall_sweep.txt
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
for cfg in "$@"; do
  s=${cfg%%:*}; p=${cfg##*:}
  echo "== slots $s plain $p" >> $OUT; SNR_PAIR_SLOTS=$s SNR_PLAIN_WGS=$p $B > /tmp/b.log 2>&1; summ /tmp/b.log >> $OUT; tail -2 /tmp/b.log | grep -i error >> $OUT
done
cat $OUT
