mkdir -p gpurun_out/r03b
(timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "roialign or split_stream or folded" 2>&1 | tail -4) > gpurun_out/r03b/tests.txt
for r in 0 1; do for v in base3 nopre nopeel nodot2 default; do
  lib=clipself_amd/csrc/ab/libclipself_hip_$v.so; [ $v = default ] && lib=clipself_amd/csrc/libclipself_hip.so
  GEMM_AB_NOREP=1 CLIPSELF_HIP_LIB=$lib python tools/gemm_ab.py 2048 1 $v 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03b/variants.txt
done; done
export TMPDIR=/tmp
root=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_inline -o r -- python $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-overlap > $root/gpurun_out/r03b/prof_inline.log 2>&1)
python tools/trace_sequence.py /tmp/prof_inline/r_results.db gpurun_out/r03b/seq_inline.txt
python tools/rocprof_summary.py /tmp/prof_inline/r_results.db "r03b inline" > gpurun_out/r03b/kernel_stats_inline.md
python tools/step_phases.py > gpurun_out/r03b/phases.txt 2>&1
cat gpurun_out/r03b/tests.txt; cat gpurun_out/r03b/variants.txt; head -60 gpurun_out/r03b/seq_inline.txt; tail -2 gpurun_out/r03b/phases.txt
