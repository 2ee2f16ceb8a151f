#!/bin/bash
# Sample the shader clock and socket power while the default training step runs (evidence for DESIGN.md §5's note that
# the step is clock / power coupled).  usage: tools/clock_sample.sh OUTFILE
OUT=${1:-gpurun_out/clock.txt}
python bench.py --steps 40000 --warmup 50 --no-cpu-baseline --no-frame --no-hashgrid > ${OUT}.bench 2>&1 &
PID=$!
sleep 12          # first torch import on a fresh box is slow
for i in $(seq 1 22); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | tr "\n" " " >> $OUT
  echo "" >> $OUT
  sleep 2.5
done
wait $PID
echo "idle:" >> $OUT
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" >> $OUT
tail -c 300 ${OUT}.bench >> $OUT
