#!/bin/bash
# Round 5, second box: parity of the changed kernels, the reference's recipe shape on the record, the ordered kernel
# sequence of an inline step, A/B of the kernarg-reload epilogue.      bash tools/r05_b.sh <tag>
tag=${1:-r05b}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "split_stream or gemm_nt_ln or stream or resid" > "$out/tests.log" 2>&1; echo "gemm tests rc $?" >> "$out/tests.log"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x >> "$out/tests.log" 2>&1; echo "parity tests rc $?" >> "$out/tests.log"
timeout 600 python -m pytest tests/test_gpu_step.py -q -x -k "mask or openai" >> "$out/tests.log" 2>&1; echo "openai tests rc $?" >> "$out/tests.log"
timeout 600 python tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 10 > "$out/recipe_b16.json" 2> "$out/recipe_b16.err"
timeout 900 python tools/recipe_bench.py EVA02-CLIP-L-14-336 896 2 6 > "$out/recipe_l14.json" 2> "$out/recipe_l14.err"
bash tools/ab_bench.sh $tag/ab 2 $root/clipself_amd/csrc/ab/libclipself_hip_nokr.so $root/clipself_amd/csrc/libclipself_hip.so > /dev/null 2>&1
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_recipe" -o r -- python $root/tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 4 > "$out/prof_recipe.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_inline" -o r -- python $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-overlap > "$out/prof_inline.log" 2>&1
cd "$root"
python tools/rocprof_summary.py "$out/prof_recipe/r_results.db" "$tag recipe shape: rocprofv3 --kernel-trace --stats -- python tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 4 (2 images x <= 20 crops at 224^2, student 1024^2 = 4097 tokens; overlapped + inline + phase runs), MI355X" > "$out/kernel_stats_recipe.md"
python tools/rocprof_summary.py "$out/prof_inline/r_results.db" "$tag inline schedule: rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap (3 steps), MI355X" > "$out/kernel_stats_inline.md"
python tools/trace_sequence.py "$out/prof_inline/r_results.db" "$out/sequence_inline.txt"
rm -rf "$out/prof_recipe" "$out/prof_inline"
cat "$out/tests.log" | tail -20; cat "$out/recipe_b16.json" "$out/recipe_l14.json"; cat "$out/ab/bench_ab.txt"
