out=gpurun_out/r03q
mkdir -p $out
AB=clipself_amd/csrc/ab
(timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or folded or split_stream or fp8" 2>&1 | tail -5) > $out/tests_ops.txt
tail -2 $out/tests_ops.txt
for r in 0 1; do
  for lib in $AB/libclipself_hip_cur.so clipself_amd/csrc/libclipself_hip.so; do
    GEMM_AB_NOREP=$( [ $r = 1 ] && echo 1 ) CLIPSELF_HIP_LIB=$lib timeout 300 python tools/gemm_ab.py 2048 1 "$(basename $lib)" 2>&1 | grep -v amdgpu.ids >> $out/gemm_ab.txt
  done
done
grep -v "0/5 runs" $out/gemm_ab.txt
grep -c "0/5 runs" $out/gemm_ab.txt
