#!/bin/bash
# Teacher side-stream priority A/B (same box, interleaved): -1 (high, default), 0 (equal), 1 (low, where the runtime has three levels)
tag=${1:-r04c}
out=gpurun_out/$tag
mkdir -p $out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" > $out/prio_range.txt 2>&1
for rep in 1 2; do
  for p in -1 0 1; do
    CLIPSELF_TEACHER_STREAM_PRIORITY=$p python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prio $p rep $rep', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', 'dominant', round(d['roofline']['mean_us'],1))" >> $out/prio_ab.txt
  done
done
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-overlap 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inline', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')" >> $out/prio_ab.txt
cat $out/prio_range.txt $out/prio_ab.txt
