#!/bin/bash
# Round 5, sixth box: statistics finalisation with 16 slices in flight (tests, micro-benchmark, step A/B against the previous library).
tag=${1:-r05f}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "finalize or stats or folded or fold or swiglu" > "$out/tests.log" 2>&1; echo "stats tests rc $?" >> "$out/tests.log"
timeout 300 python tools/small_kernel_bench.py > "$out/small_kernels.txt" 2>&1
CLIPSELF_HIP_LIB=$root/clipself_amd/csrc/ab/libclipself_hip_prev.so timeout 300 python tools/small_kernel_bench.py > "$out/small_kernels_prev.txt" 2>&1
bash tools/ab_bench.sh $tag/ab 2 $root/clipself_amd/csrc/ab/libclipself_hip_prev.so $root/clipself_amd/csrc/libclipself_hip.so > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "teacher or frozen or forward" >> "$out/tests.log" 2>&1; echo "frozen parity rc $?" >> "$out/tests.log"
tail -6 "$out/tests.log"; cat "$out/small_kernels_prev.txt" "$out/small_kernels.txt" "$out/ab/bench_ab.txt"
