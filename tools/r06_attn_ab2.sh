#!/bin/bash
# round 6: attention kernels before / after (round-5 attention.hip as ab/libclipself_hip_r5attn.so vs the current library), interleaved on one box
# usage (GPU box): bash tools/r06_attn_ab2.sh > gpurun_out/r06_d_attn_ab.txt
cd "$(dirname "$0")/.."
OLD=$(pwd)/clipself_amd/csrc/ab/libclipself_hip_r5attn.so
f() { grep -v amdgpu.ids; }
for pass in 1 2; do
  echo "== pass $pass: teacher forward, 1024 crops x 12 heads x 197 tokens"
  echo -n "r5 (fwd8, XOR V^T)        "; CLIPSELF_HIP_LIB=$OLD python tools/attn_bench.py 1024 2>&1 | f
  echo -n "r6 fwd4 packed            "; python tools/attn_bench.py 1024 2>&1 | f
  echo -n "r6 fwd8 rotated packed    "; CS_ATTN_FWD4=0 python tools/attn_bench.py 1024 2>&1 | f
  echo -n "r6 fwd8 XOR packed        "; CS_ATTN_FWD4=0 CS_ATTN_FWD8_VXOR=1 python tools/attn_bench.py 1024 2>&1 | f
  echo "== student 64 images x 12 x 197: forward + backward"
  echo "r5:"; CLIPSELF_HIP_LIB=$OLD python tools/attn_bench.py 64 bwd 2>&1 | f
  echo "r6:"; python tools/attn_bench.py 64 bwd 2>&1 | f
  echo "== recipe shape 2 x 12 x 4097"
  echo "r5:"; CLIPSELF_HIP_LIB=$OLD python tools/attn_long_bench.py 2 64 12 10 2>&1 | f
  echo "r6:"; python tools/attn_long_bench.py 2 64 12 10 2>&1 | f
  echo "== L/14-336 16 x 16 x 577"
  echo "r5:"; CLIPSELF_HIP_LIB=$OLD python tools/attn_long_bench.py 16 24 16 10 2>&1 | f
  echo "r6:"; python tools/attn_long_bench.py 16 24 16 10 2>&1 | f
done
echo "== step (bench.py --steps 12 --warmup 4 --no-cpu-baseline), interleaved"
for pass in 1 2; do
  for lib in "$OLD" ""; do
    echo -n "${lib:-current}: "
    CLIPSELF_HIP_LIB=$lib python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'images/s',round(d['ms_per_step'],2),'ms')"
  done
done
