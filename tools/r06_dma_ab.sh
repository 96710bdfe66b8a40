#!/bin/bash
cd "$(dirname "$0")/.."
root=$(pwd); out=$root/gpurun_out/r06q; mkdir -p $out
CS_ATTN_DMA=1 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention_fwd_bwd or restaged_kernels or long_sequence" 2>&1 | tail -3
export TMPDIR=/tmp; cd /tmp
for tag in base dma; do
  e=""; [ $tag = dma ] && e="CS_ATTN_DMA=1"
  env $e rocprofv3 --kernel-trace --stats -d $out/p -o r -- python $root/tools/attn_long_bench.py 2 64 12 10 > $out/p.log 2>&1
  echo -n "$tag  "; python $root/tools/rocprof_summary.py $out/p/r_results.db x | grep "attn_bwd_dq2" | cut -c1-100; rm -rf $out/p
done
for tag in base dma; do
  e=""; [ $tag = dma ] && e="CS_ATTN_DMA=1"
  env $e rocprofv3 --kernel-trace --stats -d $out/p -o r -- python $root/tools/attn_long_bench.py 16 24 16 10 > $out/p.log 2>&1
  echo -n "$tag 577 "; python $root/tools/rocprof_summary.py $out/p/r_results.db x | grep "attn_bwd_dq2" | cut -c1-100; rm -rf $out/p
done
