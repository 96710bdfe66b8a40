out=gpurun_out/r03o
mkdir -p $out
AB=clipself_amd/csrc/ab
(CLIPSELF_HIP_LIB=$AB/libclipself_hip_a8p.so timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" 2>&1 | tail -8) > $out/tests_attn.txt
tail -4 $out/tests_attn.txt
for r in 0 1; do
  for lib in clipself_amd/csrc/libclipself_hip.so $AB/libclipself_hip_a8p.so; do
    echo "# $lib" >> $out/attn.txt
    CLIPSELF_HIP_LIB=$lib timeout 120 python tools/attn_bench.py 2048 2>&1 | grep attn_fwd >> $out/attn.txt
    CLIPSELF_HIP_LIB=$lib timeout 120 python tools/attn_bench.py 64 2>&1 | grep attn_fwd >> $out/attn.txt
  done
done
cat $out/attn.txt
