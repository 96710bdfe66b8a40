#!/bin/bash
# round 6: the short-sequence attention forward's variants on the teacher's launch shape (1024 crops x 12 heads x 197 tokens), interleaved passes
# usage (GPU box): bash tools/r06_attn_ab.sh > gpurun_out/r06_b_attn_ab.txt        (profiles/r06_c_attention_pipes.md section 2 was measured with the
# round's intermediate library, which also carried a rotated V^T [64][224] image and a five-wave form of attn_fwd4_kernel: both lost and are gone)
cd "$(dirname "$0")/.."
CS_ATTN_DEBUG=1 CS_ATTN_FWD4=1 python tools/attn_bench.py 64 2>&1 | grep cs_attn | sort | uniq
CS_ATTN_DEBUG=1 python tools/attn_bench.py 64 2>&1 | grep cs_attn | sort | uniq
for pass in 1 2 3; do
  echo "pass $pass"
  echo -n "fwd8 (default: 8 waves x 2 units, XOR V^T)  "; python tools/attn_bench.py 1024 2>&1 | grep -v amdgpu
  echo -n "fwd4 (4 waves x 3 units, rotated V^T)       "; CS_ATTN_FWD4=1 python tools/attn_bench.py 1024 2>&1 | grep -v amdgpu
  echo -n "fwd8 row-major V                            "; CS_ATTN_FWD8_VROW=1 python tools/attn_bench.py 1024 2>&1 | grep -v amdgpu
done
