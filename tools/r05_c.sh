#!/bin/bash
# Round 5, third box: the re-staged attention kernels -- parity, bit-identity with the round-1 backward, micro-benchmarks at the three shapes,
# the recipe step and the headline step again.      bash tools/r05_c.sh <tag>
tag=${1:-r05c}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" > "$out/tests.log" 2>&1; echo "attention tests rc $?" >> "$out/tests.log"
timeout 600 python tools/attn_long_bench.py 2 64 12 > "$out/attn_bench.txt" 2>&1
timeout 600 python tools/attn_long_bench.py 64 14 12 20 >> "$out/attn_bench.txt" 2>&1
timeout 600 python tools/attn_long_bench.py 16 24 16 >> "$out/attn_bench.txt" 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "recipe or multiscale or reproducible" >> "$out/tests.log" 2>&1; echo "fullsize tests rc $?" >> "$out/tests.log"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py -q -x >> "$out/tests.log" 2>&1; echo "parity + step tests rc $?" >> "$out/tests.log"
timeout 600 python tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 10 > "$out/recipe_b16.json" 2> "$out/recipe_b16.err"
timeout 900 python tools/recipe_bench.py EVA02-CLIP-L-14-336 896 2 6 > "$out/recipe_l14.json" 2> "$out/recipe_l14.err"
for r in 1 2; do
  CS_ATTN_BWD_V1=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bwd v1 rep $r: %.1f images/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> "$out/bench_ab.txt"
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bwd v2 rep $r: %.1f images/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> "$out/bench_ab.txt"
done
tail -12 "$out/tests.log"; cat "$out/attn_bench.txt" "$out/recipe_b16.json" "$out/recipe_l14.json" "$out/bench_ab.txt"
