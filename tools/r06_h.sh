#!/bin/bash
cd "$(dirname "$0")/.."
V=$(pwd)/clipself_amd/csrc/ab/libclipself_hip_dqfull.so
root=$(pwd); out=$root/gpurun_out/r06h; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
for tag in base dqfull; do
  lib=""; [ $tag = dqfull ] && lib=$V
  CLIPSELF_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d $out/p_$tag -o r -- python $root/tools/attn_long_bench.py 2 64 12 10 > $out/p_$tag.log 2>&1
  python $root/tools/rocprof_summary.py $out/p_$tag/r_results.db "$tag" | grep "attn_" ; rm -rf $out/p_$tag
done
cd $root
for pass in 1 2; do for tag in base dqfull; do lib=""; [ $tag = dqfull ] && lib=$V
  echo -n "$tag: "; CLIPSELF_HIP_LIB=$lib python tools/attn_long_bench.py 2 64 12 10 2>&1 | grep "v2 pass 1" | cut -c40-200
done; done
