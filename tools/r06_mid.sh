#!/bin/bash
# Round 6, mid-round evidence: the headline step's launch sequence (inline schedule), kernel stats of the headline and of the recipe shape.
tag=${1:-r06m}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout 300 python tools/step_phases.py > "$out/phases.txt" 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_seq" -o r -- python $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-overlap > "$out/prof_seq.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_recipe" -o r -- python $root/tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 4 > "$out/prof_recipe.log" 2>&1
cd "$root"
python tools/trace_sequence.py "$out/prof_seq/r_results.db" "$out/sequence_inline.txt"
python tools/rocprof_summary.py "$out/prof_seq/r_results.db" "$tag inline schedule: rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap (3 steps), MI355X" > "$out/kernel_stats_inline.md"
python tools/rocprof_summary.py "$out/prof_recipe/r_results.db" "$tag recipe shape: rocprofv3 --kernel-trace --stats -- python tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 4, MI355X" > "$out/kernel_stats_recipe.md"
python tools/trace_sequence.py "$out/prof_recipe/r_results.db" "$out/sequence_recipe.txt"
rm -rf "$out/prof_seq" "$out/prof_recipe"
grep -h '^{' "$out/prof_recipe.log" > "$out/recipe_under_profiler.json"
cat "$out/phases.txt"; head -30 "$out/kernel_stats_recipe.md"
