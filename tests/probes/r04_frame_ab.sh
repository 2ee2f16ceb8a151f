#!/bin/bash
# frame-time A/B of inference-kernel variants (SNR_LIB), same call
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset SNR_LIB; else export SNR_LIB=$PWD/spin-nerf_amd/lib/ablate/libspinnerf_hip_$v.so; fi
  python bench.py --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --no-hashgrid 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', 'frame %.2f ms' % d['ms_per_frame_378x504'], 'step %.4f' % d['ms_per_step'])"
done; done
