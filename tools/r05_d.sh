#!/bin/bash
# Round 5, fourth box: the backward kernels after the VALU trim (no masks, raw v_exp_f32, folded scale, tiles unrolled).   bash tools/r05_d.sh <tag>
tag=${1:-r05d}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" > "$out/tests.log" 2>&1; echo "attention tests rc $?" >> "$out/tests.log"
grep "attn_bwd v2 vs v1" gpurun_out/ops_metrics.txt | tail -18 > "$out/v2_vs_v1.txt"
timeout 600 python tools/attn_long_bench.py 2 64 12 > "$out/attn_bench.txt" 2>&1
timeout 600 python tools/attn_long_bench.py 64 14 12 20 >> "$out/attn_bench.txt" 2>&1
timeout 600 python tools/attn_long_bench.py 16 24 16 >> "$out/attn_bench.txt" 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "recipe or reproducible or inline_and_prefetch" >> "$out/tests.log" 2>&1; echo "fullsize tests rc $?" >> "$out/tests.log"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "backward" >> "$out/tests.log" 2>&1; echo "block backward parity rc $?" >> "$out/tests.log"
timeout 600 python tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 10 > "$out/recipe_b16.json" 2> "$out/recipe_b16.err"
for r in 1 2; do
  CS_ATTN_BWD_V1=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bwd v1 rep $r: %.1f images/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> "$out/bench_ab.txt"
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bwd v2 rep $r: %.1f images/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> "$out/bench_ab.txt"
done
tail -8 "$out/tests.log"; cat "$out/v2_vs_v1.txt" "$out/attn_bench.txt" "$out/recipe_b16.json" "$out/bench_ab.txt"
