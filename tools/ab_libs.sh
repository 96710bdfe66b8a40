#!/bin/bash
# Interleaved A/B of two builds of libclipself_hip.so on one box:  bash tools/ab_libs.sh <tag> <libA> <libB> [crops]
#   -> gpurun_out/<tag>/gemm_ab.txt (tools/gemm_ab.py per library, twice, interleaved) and bench_ab.jsonl (bench.py per library, twice)
tag=$1; A=$2; B=$3; crops=${4:-2048}
out=gpurun_out/$tag
mkdir -p $out
for r in 0 1; do
  for lib in $A $B; do
    CLIPSELF_HIP_LIB=$lib python tools/gemm_ab.py $crops 1 "$(basename $(dirname $lib))/$(basename $lib)" 2>&1 | grep -v amdgpu.ids >> $out/gemm_ab.txt
  done
done
for r in 0 1; do
  for lib in $A $B; do
    echo "# $lib" >> $out/bench_ab.jsonl
    CLIPSELF_HIP_LIB=$lib python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{' >> $out/bench_ab.jsonl
  done
done
cat $out/gemm_ab.txt
python - $out/bench_ab.jsonl <<'PY'
import json, sys
lib = None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        lib = line[2:].strip()
    else:
        d = json.loads(line)
        print(f"{lib}: {d['value']:.1f} images/s, {d['ms_per_step']:.2f} ms/step, dominant kernel {d['roofline']['mean_us']:.0f} us")
PY
