#!/bin/bash
# Per-round profile evidence on the GPU box (rocprofv3; PMC counters in their own passes, never with a trace domain):
#   bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/{kernel_stats_inline,kernel_stats_overlap,pmc_traffic,pmc_mfma}.md + pmc_traffic.json
# Copy the summaries to profiles/ (tracked) afterwards.
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
B="python $root/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$out/prof_inline" -o r -- $B --steps 2 --warmup 1 --no-overlap > "$out/prof_inline.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/prof_overlap" -o r -- $B --steps 5 --warmup 2 > "$out/prof_overlap.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_f" -o r -- $B --steps 1 --warmup 1 --no-overlap > "$out/pmc_f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_w" -o r -- $B --steps 1 --warmup 1 --no-overlap > "$out/pmc_w.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAIT_ANY -d "$out/pmc_m" -o r -- $B --steps 1 --warmup 1 --no-overlap > "$out/pmc_m.log" 2>&1
cd "$root"
python tools/rocprof_summary.py "$out/prof_inline/r_results.db" "$tag inline schedule: rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap (3 steps), MI355X" > "$out/kernel_stats_inline.md"
python tools/rocprof_summary.py "$out/prof_overlap/r_results.db" "$tag default (overlapped) schedule: rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline (7 steps), MI355X" > "$out/kernel_stats_overlap.md"
python tools/rocprof_pmc.py --dominant "gemm_stream_kernel<3" --chunk 2048 --out "$out/pmc_traffic.json" "$out/pmc_f/r_results.db" "$out/pmc_w/r_results.db" > "$out/pmc_traffic.md"
python tools/rocprof_pmc.py "$out/pmc_m/r_results.db" > "$out/pmc_mfma.md"
grep -h '^{' "$out/prof_inline.log" "$out/prof_overlap.log" > "$out/bench_under_profiler.jsonl"
rm -rf "$out"/prof_inline "$out"/prof_overlap "$out"/pmc_f "$out"/pmc_w "$out"/pmc_m        # keep the summaries, not the raw databases
ls -la "$out"
