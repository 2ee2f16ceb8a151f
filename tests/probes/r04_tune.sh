#!/bin/bash
# Round 4: tuning of the merged weight-gradient launch on one box: slot weight of the encoding-layer pairs (SNR_PAIR_W0), pair
# slots vs plain workgroups, pacing.   args: "W0:slots:plain[:poll:lead]" ...
OUT=gpurun_out/r04_tune.txt
mkdir -p gpurun_out; : > $OUT
B="python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-hashgrid --no-frame"
summ() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); k = d["kernels"]
        print("step %.4f  " % d["ms_per_step"] + "  ".join("%s %.4f" % (n.replace("mlp_", ""), k[n]["ms_per_step"]) for n in ("mlp_fwd", "mlp_dgrad", "mlp_wgrad_pair", "mlp_wgrad_reduce", "adam", "composite_fwd", "make_rays") if n in k))
PY
}
for cfg in "$@"; do
  IFS=: read w s p poll lead <<< "$cfg"
  echo "== W0 $w slots $s plain $p poll ${poll:-16} lead ${lead:-2}" >> $OUT
  SNR_PAIR_W0=$w SNR_PAIR_SLOTS=$s SNR_PLAIN_WGS=$p SNR_PAIR_POLL=${poll:-16} SNR_PAIR_LEAD=${lead:-2} $B > /tmp/b.log 2>&1; summ /tmp/b.log >> $OUT
done
cat $OUT
