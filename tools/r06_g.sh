#!/bin/bash
# round 6: backward kernels with a branch-free full-chunk path (default library) vs per-tile exits (ab/libclipself_hip_nofull.so), interleaved
cd "$(dirname "$0")/.."
OLD=$(pwd)/clipself_amd/csrc/ab/libclipself_hip_nofull.so
f() { grep -v amdgpu.ids | grep "v2 pass"; }
for shape in "2 64 12" "16 24 16" "4 32 12"; do
  for pass in 1 2; do
    echo "== $shape pass $pass"
    echo -n "per-tile exits : "; CLIPSELF_HIP_LIB=$OLD python tools/attn_long_bench.py $shape 10 2>&1 | f | tail -1
    echo -n "full path      : "; python tools/attn_long_bench.py $shape 10 2>&1 | f | tail -1
  done
done
