#!/bin/bash
# Round-4 diagnostics on the GPU box: (1) bench line, (2) phase times, (3) the ordered launch sequence of one inline step and one overlapped step
# (stray fills / copies / gaps), (4) an LDS PMC pass on the step (bank conflicts vs LDS instructions vs LDS-array cycles of the tower GEMMs).
#   bash tools/r04_diag.sh <tag>   -> gpurun_out/<tag>/
tag=${1:-r04a}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python $root/bench.py --no-cpu-baseline"
$B --steps 10 --warmup 3 > "$out/bench.json" 2> "$out/bench.err"
python tools/step_phases.py > "$out/phases.txt" 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d "$out/prof_inline" -o r -- $B --steps 2 --warmup 1 --no-overlap > "$out/prof_inline.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/prof_overlap" -o r -- $B --steps 3 --warmup 1 > "$out/prof_overlap.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$out/pmc_lds" -o r -- $B --steps 1 --warmup 1 --no-overlap > "$out/pmc_lds.log" 2>&1
cd "$root"
python tools/rocprof_summary.py "$out/prof_inline/r_results.db" "$tag inline" > "$out/kernel_stats_inline.md"
python tools/rocprof_summary.py "$out/prof_overlap/r_results.db" "$tag overlap" > "$out/kernel_stats_overlap.md"
python tools/trace_sequence.py "$out/prof_inline/r_results.db" "$out/sequence_inline.txt"
python tools/trace_sequence.py "$out/prof_overlap/r_results.db" "$out/sequence_overlap.txt"
python tools/rocprof_pmc.py "$out/pmc_lds/r_results.db" > "$out/pmc_lds.md"
rm -rf "$out"/prof_inline "$out"/prof_overlap "$out"/pmc_lds
ls -la "$out"
