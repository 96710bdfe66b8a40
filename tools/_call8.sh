out=gpurun_out/r03h
mkdir -p $out
AB=clipself_amd/csrc/ab
for r in 0 1; do
  for v in cur_abl cur_nodma new0_abl new_nodma; do
    ABLATE_DBG=4,6,4,6 CLIPSELF_HIP_LIB=$AB/libclipself_hip_$v.so timeout 300 python tools/stream_ablate.py 2048 $v 2>&1 | grep -v amdgpu.ids >> $out/ablate.txt
  done
done
cat $out/ablate.txt
