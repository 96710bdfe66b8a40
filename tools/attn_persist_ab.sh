#!/bin/bash
# A/B of the attention forward's unit loop: CS_ATTN_WGS=0 (one workgroup per unit, rounds 1-3) / 512 (two per CU walk the units)
for r in 1 2 3; do
  for w in 0 512; do
    echo -n "CS_ATTN_WGS=$w  "; CS_ATTN_WGS=$w python tools/attn_bench.py 1024 2>&1 | tail -1
  done
done
export CLIPSELF_HIP_LIB=clipself_amd/csrc/ab/libclipself_hip_abl.so
for w in 0 512; do
  echo "== timeline, CS_ATTN_WGS=$w"
  CS_ATTN_WGS=$w CS_ATTN_TRACE=gpurun_out/attn_trace_$w.bin python tools/attn_bench.py 1024 > /dev/null 2>&1 && python tools/attn_trace.py gpurun_out/attn_trace_$w.bin
done
