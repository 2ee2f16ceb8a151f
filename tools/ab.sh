#!/bin/bash
# A/B of library variants inside ONE gpurun call (box-to-box variance is +-2-3 %):  tools/ab.sh OUTDIR name1 name2 ...
# "base" = the in-tree library; other names = spin-nerf_amd/lib/ablate/libspinnerf_hip_<name>.so (tools/build_variant.py)
OUT=$1; shift
mkdir -p $OUT
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
    python bench.py --no-cpu-baseline --no-hashgrid > $OUT/${v}_$rep.log 2>&1
  done
done
python - "$OUT" "$@" <<'PY'
import json, sys, glob
out = sys.argv[1]
for v in sys.argv[2:]:
    for f in sorted(glob.glob(f"{out}/{v}_*.log")):
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l); k = d["kernels"]
                print(f"{v:10s} step {d['ms_per_step']:.4f} ms  frame {d.get('ms_per_frame_378x504') or 0:.2f} ms  fwd {k['mlp_fwd']['ms_per_step']:.4f} dgrad {k['mlp_dgrad']['ms_per_step']:.4f} wgrad {k['mlp_wgrad']['ms_per_step']:.4f}")
PY
