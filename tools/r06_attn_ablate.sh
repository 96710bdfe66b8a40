#!/bin/bash
# round 6: what the long-sequence dQ kernel's time is made of (ablation build: results wrong by construction), 2 x 12 x 4097 tokens
cd "$(dirname "$0")/.."
L=$(pwd)/clipself_amd/csrc/ab/libclipself_hip_abl.so
root=$(pwd); out=$root/gpurun_out/r06p; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
for dbg in 0 16 32 48 64 80 96 112; do
  CLIPSELF_HIP_LIB=$L CS_ATTN_DBG=$dbg rocprofv3 --kernel-trace --stats -d $out/p -o r -- python $root/tools/attn_long_bench.py 2 64 12 5 > $out/p.log 2>&1
  echo -n "CS_ATTN_DBG=$dbg  "; python $root/tools/rocprof_summary.py $out/p/r_results.db x | grep "attn_bwd_dq2" | cut -c1-90; rm -rf $out/p
done
