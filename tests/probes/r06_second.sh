#!/bin/bash
# second GPU call of round 6: the fp16-encoding build — full suite (poisoned, full output), the bf16 error measurements on the
# reference fixtures for both arithmetic variants, a same-call A/B of the two libraries, and a re-sweep of the merged weight-gradient
# launch's split (pair slots / plain workgroups / slot weight of pair 0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
SNR_POISON_WS=1 timeout 1500 python -m pytest tests -q -m gpu --tb=long -rA -p no:cacheprovider > $OUT/r06_second_suite.txt 2>&1
echo "suite exit=$? : $(tail -1 $OUT/r06_second_suite.txt)"; grep "^FAILED\|^ERROR" $OUT/r06_second_suite.txt | head -30
for v in base bf16enc; do
  if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  timeout 600 python tests/probes/trained_diag.py > $OUT/r06_trained_diag_$v.txt 2>&1
  timeout 600 python tests/probes/render_diag.py > $OUT/r06_render_diag_$v.txt 2>&1
done
unset SNR_LIB
bash tools/ab.sh $OUT/r06_ab_enc base bf16enc > $OUT/r06_ab_enc.txt 2>&1; cat $OUT/r06_ab_enc.txt
: > $OUT/r06_tune.txt
B="python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-hashgrid --no-frame"
for cfg in 80:108:40 80:104:48 80:112:32 80:116:24 80:100:56 70:108:40 90:108:40 100:108:40 80:108:40:16:1 80:108:40:16:3 80:108:40:8:2 80:110:36 80:106:44; do
  IFS=: read w s p poll lead <<< "$cfg"
  SNR_PAIR_W0=$w SNR_PAIR_SLOTS=$s SNR_PLAIN_WGS=$p SNR_PAIR_POLL=${poll:-16} SNR_PAIR_LEAD=${lead:-2} $B > /tmp/b.log 2>&1
  python - "$cfg" >> $OUT/r06_tune.txt <<'PY'
import json, sys
for l in open("/tmp/b.log"):
    if l.startswith("{"):
        d = json.loads(l); k = d["kernels"]
        print("W0:slots:plain[:poll:lead] %-16s step %.4f  " % (sys.argv[1], d["ms_per_step"]) + "  ".join("%s %.4f" % (n.replace("mlp_", ""), k[n]["ms_per_step"]) for n in ("mlp_fwd", "mlp_dgrad", "mlp_wgrad_pair", "mlp_wgrad_reduce") if n in k))
PY
done
cat $OUT/r06_tune.txt
