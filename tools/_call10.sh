out=gpurun_out/r03j
mkdir -p $out
AB=clipself_amd/csrc/ab
(CLIPSELF_HIP_LIB=$AB/libclipself_hip_s4.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or folded or split_stream" 2>&1 | tail -15) > $out/tests_ops.txt
tail -12 $out/tests_ops.txt
for r in 0 1; do
  for lib in clipself_amd/csrc/libclipself_hip.so $AB/libclipself_hip_s4.so; do
    GEMM_AB_NOREP=$( [ $r = 1 ] && echo 1 ) CLIPSELF_HIP_LIB=$lib timeout 300 python tools/gemm_ab.py 2048 1 "$(basename $lib)" 2>&1 | grep -v amdgpu.ids >> $out/gemm_ab.txt
  done
done
cat $out/gemm_ab.txt
