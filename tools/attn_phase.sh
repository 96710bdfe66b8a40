#!/bin/bash
# Start-phase experiment of the attention forward (ablation build): bash tools/attn_phase.sh  -> stdout
export CLIPSELF_HIP_LIB=clipself_amd/csrc/ab/libclipself_hip_abl.so
for d in 0 256 512 768 1024 1536 2048 $((64+512)) $((64+1024)) $((128+512)) $((128+1024)) $((128+2048)) 0; do
  echo -n "CS_ATTN_DBG=$d  "; CS_ATTN_DBG=$d python tools/attn_bench.py 1024 2>&1 | tail -1
done
for d in 1 4 5; do
  echo -n "CS_ATTN_DBG=$d (1 = no MFMA/softmax, 4 = no stores)  "; CS_ATTN_DBG=$d python tools/attn_bench.py 1024 2>&1 | tail -1
done
