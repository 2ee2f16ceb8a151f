#!/bin/bash
# GPU call: full -m gpu suite (no -x: every failure is wanted), the render-gate probe, the bench line
OUT=gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -q --timeout 1200 2>&1 | tail -60 > $OUT/c1_tests.log
python tests/probes/render_diag.py > $OUT/c1_render_diag.jsonl 2> $OUT/c1_render_diag.err
python bench.py --steps 20 --warmup 5 > $OUT/c1_bench.json 2> $OUT/c1_bench.err
tail -5 $OUT/c1_tests.log; tail -c 1500 $OUT/c1_bench.json
