out=gpurun_out/r03n
mkdir -p $out
AB=clipself_amd/csrc/ab
for d in 0 1 2 4 3 7 0; do
  echo "# CS_ATTN_DBG=$d" >> $out/attn_abl.txt
  CS_ATTN_DBG=$d CS_ATTN_DEBUG=1 CLIPSELF_HIP_LIB=$AB/libclipself_hip_a8abl.so timeout 120 python tools/attn_bench.py 2048 2>&1 | grep "attn_fwd\|resident" | sort -u >> $out/attn_abl.txt
done
cat $out/attn_abl.txt
