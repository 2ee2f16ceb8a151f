#!/bin/bash
# chain-kernel grid size (SNR_CHAIN_GRID): 1024 (default: blocks queue behind each other) vs persistent 256 / 512
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for g in 0 256 512; do
  SNR_CHAIN_GRID=$g python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-hashgrid 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('grid $g', 'step %.4f' % d['ms_per_step'], 'frame %.2f' % (d.get('ms_per_frame_378x504') or 0), {n: round(k[n]['ms_per_step'],4) for n in ('mlp_fwd','mlp_dgrad','mlp_wgrad_pair')})"
done; done
