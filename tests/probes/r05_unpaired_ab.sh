#!/bin/bash
# A/B of the layer-pair launch: one workgroup of each kind per slot (shipped) against workgroups per (pair, kind) by weight.
# Needs the experiment's kernel (commit c990628, `git show c990628:spin-nerf_amd/csrc/mlp_wgrad_pair.h`): the switch was
# removed again after it measured nothing (profiles/r05_pair_unpaired_ab.txt).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SNR_PAIR_UNPAIRED=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py -m gpu -x -q -k "backward or wgrad or gradient or step" 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 5 --blocks 3 --no-cpu-baseline --no-frame --no-hashgrid"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']
        print('$1', 'step %.4f' % d['ms_per_step'], 'pair %.4f' % k['mlp_wgrad_pair']['ms_per_step'], 'fwd %.4f' % k['mlp_fwd']['ms_per_step'], 'dgrad %.4f' % k['mlp_dgrad']['ms_per_step'], 'reduce %.4f' % k['mlp_wgrad_reduce']['ms_per_step'])
"; }
for rep in 1 2; do
  $B 2>/dev/null | line "paired            "
  SNR_PAIR_UNPAIRED=1 $B 2>/dev/null | line "unpaired 93/76/84 "
  SNR_PAIR_UNPAIRED=1 SNR_PAIR_WA=100 SNR_PAIR_W0A=80 SNR_PAIR_W0B=80 $B 2>/dev/null | line "unpaired 100/80/80"
  SNR_PAIR_UNPAIRED=1 SNR_PAIR_WA=88 SNR_PAIR_W0A=72 SNR_PAIR_W0B=84 $B 2>/dev/null | line "unpaired 88/72/84 "
  SNR_PAIR_UNPAIRED=1 SNR_PAIR_WA=96 SNR_PAIR_W0A=78 SNR_PAIR_W0B=82 $B 2>/dev/null | line "unpaired 96/78/82 "
done
