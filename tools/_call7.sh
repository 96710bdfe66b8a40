out=gpurun_out/r03g
mkdir -p $out
AB=clipself_amd/csrc/ab
for r in 0 1; do
  for v in cur_abl new0_abl new1_abl new2_abl; do
    CLIPSELF_HIP_LIB=$AB/libclipself_hip_$v.so timeout 300 python tools/stream_ablate.py 2048 $v 2>&1 | grep -v amdgpu.ids >> $out/ablate.txt
  done
done
cat $out/ablate.txt
