#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r06n; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/p4 -o r -- python $root/tools/also_bench.py cfg4_bf16 4 > $out/p4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/p3 -o r -- python $root/tools/also_bench.py cfg3 3 > $out/p3.log 2>&1
cd $root
python tools/rocprof_summary.py $out/p4/r_results.db "cfg4 bf16 (L/14-336 RegionCLIP, 32 images): rocprofv3 --kernel-trace --stats -- python tools/also_bench.py cfg4_bf16 4" > $out/kernel_stats_cfg4.md
python tools/rocprof_summary.py $out/p3/r_results.db "cfg3 (L/14-336 CLIPSelf, 16 x 32 crops): rocprofv3 --kernel-trace --stats -- python tools/also_bench.py cfg3 3" > $out/kernel_stats_cfg3.md
rm -rf $out/p4 $out/p3
head -32 $out/kernel_stats_cfg4.md; head -24 $out/kernel_stats_cfg3.md
