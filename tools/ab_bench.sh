#!/bin/bash
# Interleaved bench.py A/B of several library builds on one box:  bash tools/ab_bench.sh <tag> <reps> <lib> [<lib> ...]
tag=$1; reps=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    CLIPSELF_HIP_LIB=$lib python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib rep $r: %.1f images/s %.2f ms/step dominant %.0f us' % (d['value'], d['ms_per_step'], d['roofline']['mean_us']))" >> $out/bench_ab.txt
  done
done
cat $out/bench_ab.txt
