out=gpurun_out/r03r
mkdir -p $out
AB=clipself_amd/csrc/ab
for r in 0 1; do
  for v in cur dp2x3 dp4x3 dp4x2 dp8x1; do
    GEMM_AB_NOREP=1 CLIPSELF_HIP_LIB=$AB/libclipself_hip_$v.so timeout 300 python tools/gemm_ab.py 2048 1 "$v" 2>&1 | grep -v amdgpu.ids >> $out/gemm_ab.txt
  done
done
cat $out/gemm_ab.txt
