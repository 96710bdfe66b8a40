#!/bin/bash
# round 6: the short-sequence attention forward's variants on the teacher's launch shape (1024 crops x 12 heads x 197 tokens), interleaved passes
# usage (GPU box): bash tools/r06_attn_ab.sh > gpurun_out/r06_b_attn_ab.txt
cd "$(dirname "$0")/.."
CS_ATTN_DEBUG=1 python tools/attn_bench.py 64 2>&1 | grep cs_attn | sort | uniq
CS_ATTN_DEBUG=1 CS_ATTN_FWD4=0 python tools/attn_bench.py 64 2>&1 | grep cs_attn | sort | uniq
for pass in 1 2 3; do
  echo "pass $pass"
  echo -n "fwd4 (4 waves x 3 units)       "; python tools/attn_bench.py 1024
  echo -n "fwd4 (5 waves x 3 units)       "; CS_ATTN_FWD4=5 python tools/attn_bench.py 1024
  echo -n "fwd8 rotated V^T [64][224]     "; CS_ATTN_FWD4=0 python tools/attn_bench.py 1024
  echo -n "fwd8 XOR V^T [64][264] (r1-5)  "; CS_ATTN_FWD4=0 CS_ATTN_FWD8_VXOR=1 python tools/attn_bench.py 1024
  echo -n "fwd8 row-major V               "; CS_ATTN_FWD4=0 CS_ATTN_FWD8_VROW=1 python tools/attn_bench.py 1024
done
