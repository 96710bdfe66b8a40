#!/bin/bash
# Round 5: SwiGLU backward with fused bias gradients -- tests, phases, step A/B against the previous library.
tag=${1:-r05i}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "swiglu or colsum or attention_forward_row_major" > "$out/tests.log" 2>&1; echo "ops tests rc $?" >> "$out/tests.log"
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py -q -x >> "$out/tests.log" 2>&1; echo "step + parity rc $?" >> "$out/tests.log"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k "reproducible or cfg3 or cfg4_l14_336_regionclip_real" >> "$out/tests.log" 2>&1; echo "fullsize rc $?" >> "$out/tests.log"
timeout 300 python tools/step_phases.py > "$out/phases.txt" 2>&1
mkdir -p "$out/ab"
for r in 1 2 3; do
  CLIPSELF_NO_FUSED_SWIGLU_COLSUM=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('separate colsum rep $r: %.1f images/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> "$out/ab/bench_ab.txt"
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fused colsum rep $r: %.1f images/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> "$out/ab/bench_ab.txt"
done
CLIPSELF_NO_FUSED_SWIGLU_COLSUM=1 timeout 300 python tools/step_phases.py >> "$out/phases.txt" 2>&1
tail -8 "$out/tests.log"; cat "$out/phases.txt" "$out/ab/bench_ab.txt"
