#!/bin/bash
# Round 5, final evidence on one box: tests touched by the last changes, the recipe step, the round's profile set (tools/profile_round.sh), the default bench line.
tag=${1:-r05z}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "roi or cosine or finalize" > "$out/tests.log" 2>&1; echo "roi tests rc $?" >> "$out/tests.log"
timeout 900 python -m pytest tests/test_gpu_step.py -q -x >> "$out/tests.log" 2>&1; echo "step tests rc $?" >> "$out/tests.log"
timeout 600 python tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 10 > "$out/recipe_b16.json" 2> "$out/recipe_b16.err"
timeout 900 python tools/recipe_bench.py EVA02-CLIP-L-14-336 896 2 6 > "$out/recipe_l14.json" 2> "$out/recipe_l14.err"
timeout 300 python tools/step_phases.py > "$out/phases.txt" 2>&1
bash tools/profile_round.sh $tag > "$out/profile_round.log" 2>&1
timeout 900 python bench.py > "$out/bench.json" 2> "$out/bench.err"
tail -5 "$out/tests.log"; cat "$out/recipe_b16.json" "$out/recipe_l14.json" "$out/phases.txt" "$out/bench.json"
