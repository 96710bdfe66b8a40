#!/bin/bash
# Round 6, evidence at the final commit: smoke, the round's profile set (tools/profile_round.sh), phases, recipe step, the default bench line.
tag=${1:-r06z}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout 600 python __graft_entry__.py smoke > "$out/smoke.txt" 2>&1; echo "smoke rc $?" >> "$out/smoke.txt"
timeout 300 python tools/step_phases.py > "$out/phases.txt" 2>&1
timeout 600 python tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 10 > "$out/recipe_b16.json" 2> "$out/recipe_b16.err"
bash tools/profile_round.sh $tag > "$out/profile_round.log" 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_seq" -o r -- python $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-overlap > "$out/prof_seq.log" 2>&1
cd "$root"
python tools/trace_sequence.py "$out/prof_seq/r_results.db" "$out/sequence_inline.txt"; rm -rf "$out/prof_seq"
timeout 900 python bench.py > "$out/bench.json" 2> "$out/bench.err"
tail -3 "$out/smoke.txt"; cat "$out/phases.txt" "$out/recipe_b16.json" "$out/bench.json"; head -3 "$out/sequence_inline.txt"
