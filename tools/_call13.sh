out=gpurun_out/r03m
mkdir -p $out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -6) > $out/tests.txt
tail -4 $out/tests.txt
cp gpurun_out/parity_metrics.txt $out/ 2>/dev/null
for r in 0 1; do
  for f in 0 0x1000; do
    GEMM_AB_NOREP=1 GEMM_AB_FLAGS=$f timeout 300 python tools/gemm_ab.py 2048 1 "flags=$f" 2>&1 | grep -v amdgpu.ids >> $out/gemm_ab.txt
  done
done
cat $out/gemm_ab.txt
