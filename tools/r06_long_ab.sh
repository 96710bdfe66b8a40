#!/bin/bash
# round 6: long-sequence attention variants (read per launch), interleaved on one box:
#   CS_ATTN_NOSPLIT=1 rectangular grid (round 5) | CS_ATTN_NOPRE=1 every row block rotates its own operands (round 5) | default: tail split + rotated q|k prepass
cd "$(dirname "$0")/.."
f() { grep -v amdgpu.ids | grep "v2 pass"; }
for shape in "2 64 12" "2 64 16" "16 24 16" "2 28 12" "4 32 12"; do
  for pass in 1 2; do
    echo "== $shape pass $pass"
    echo -n "round 5 (rectangular, no prepass) : "; CS_ATTN_NOSPLIT=1 CS_ATTN_NOPRE=1 python tools/attn_long_bench.py $shape 10 2>&1 | f | tail -1
    echo -n "tail split only                   : "; CS_ATTN_NOPRE=1 python tools/attn_long_bench.py $shape 10 2>&1 | f | tail -1
    echo -n "default (split + prepass)         : "; python tools/attn_long_bench.py $shape 10 2>&1 | f | tail -1
  done
done
