#!/bin/bash
# Round 4: one weight-gradient launch for both networks (SNR_MERGE_NETS=1, default) against one per network (=0), pair slots
# alone (debug build, SNR_PAIR_KIND=3) and the full launch, same box.   args: kind:merge:slots:plain ...
OUT=gpurun_out/r04_merge_ab.txt
mkdir -p gpurun_out; : > $OUT
export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_pairdbg.so
B="python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-hashgrid --no-frame"
summ() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); k = d["kernels"]
        print("step %.4f  " % d["ms_per_step"] + "  ".join("%s %.4f x%d" % (n.replace("mlp_", ""), k[n]["ms_per_step"], k[n]["launches_per_step"]) for n in ("mlp_fwd", "mlp_dgrad", "mlp_wgrad_pair", "mlp_wgrad", "mlp_wgrad_reduce") if n in k))
PY
}
for cfg in "$@"; do
  IFS=: read k m s p <<< "$cfg"
  echo "== kind $k merge $m slots $s plain $p" >> $OUT; SNR_PAIR_KIND=$k SNR_MERGE_NETS=$m SNR_PAIR_SLOTS=$s SNR_PLAIN_WGS=$p $B > /tmp/b.log 2>&1; summ /tmp/b.log >> $OUT
done
cat $OUT
