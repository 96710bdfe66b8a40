#!/bin/bash
# Round 5, VERDICT item 1 on one box: parity tests of the CU partition, the sweep, bench.py A/B of the best share against the shared pool,
# rocprofv3 kernel stats of the partitioned schedule.   bash tools/r05_partition.sh <tag> [sweep configs...]
tag=${1:-r05a}; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
confs=${@:-0 16 24 32 40 48 64 24:32 32:48 inline 32m 48m}
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "raster_modes" > "$out/tests.log" 2>&1; echo "raster tests rc $?" >> "$out/tests.log"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "partition and False" >> "$out/tests.log" 2>&1; echo "partition test (grids only) rc $?" >> "$out/tests.log"
timeout 900 python tools/partition_sweep.py 12 2 $confs > "$out/sweep.jsonl" 2> "$out/sweep.err"; echo "sweep rc $?" >> "$out/tests.log"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k "partition and True" >> "$out/tests.log" 2>&1; echo "partition test (CU masks) rc $?" >> "$out/tests.log"
best=$(python - "$out/sweep.jsonl" <<'PY'
import json, sys, collections
acc = collections.defaultdict(list)
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        if d["config"].isdigit():
            acc[d["config"]].append(d["images_per_s"])
best = max(acc, key=lambda k: sum(acc[k]) / len(acc[k])) if acc else "0"
print(best)
PY
)
echo "best share: $best" >> "$out/tests.log"
for rep in 1 2; do
  for r in 0 $best; do
    CLIPSELF_PARTITION_CUS=$r timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('partition $r rep $rep: %.1f images/s %.2f ms/step dominant %.0f us (%.3f of peak)' % (d['value'], d['ms_per_step'], d['roofline']['mean_us'], d['roofline']['frac']))" >> "$out/bench_ab.txt"
  done
done
export TMPDIR=/tmp
cd /tmp
CLIPSELF_PARTITION_CUS=$best timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof" -o r -- python $root/bench.py --no-cpu-baseline --steps 5 --warmup 2 > "$out/prof.log" 2>&1
cd "$root"
python tools/rocprof_summary.py "$out/prof/r_results.db" "$tag partitioned schedule (teacher leaves $best CUs, student capped at $best): rocprofv3 --kernel-trace --stats -- CLIPSELF_PARTITION_CUS=$best python bench.py --steps 5 --warmup 2 --no-cpu-baseline (7 steps), MI355X" > "$out/kernel_stats_partition.md"
grep -h '^{' "$out/prof.log" > "$out/bench_under_profiler.jsonl"
rm -rf "$out/prof"
cat "$out/tests.log" "$out/bench_ab.txt"; cat "$out/sweep.jsonl"
