#!/bin/bash
# Round 4: the plain workgroups of the merged weight-gradient launch ALONE (debug build: SNR_PAIR_KIND=2) and the pair slots
# alone (SNR_PAIR_KIND=3), against the full launch — does the other role slow them down?
OUT=gpurun_out/r04_plain_alone.txt
mkdir -p gpurun_out; : > $OUT
export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_pairdbg.so
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
  k=${cfg%%:*}; r=${cfg#*:}; s=${r%%:*}; p=${r##*:}
  echo "== kind $k slots $s plain $p" >> $OUT; SNR_PAIR_KIND=$k SNR_PAIR_SLOTS=$s SNR_PLAIN_WGS=$p $B > /tmp/b.log 2>&1; summ /tmp/b.log >> $OUT
done
cat $OUT
