out=gpurun_out/r03f
mkdir -p $out
AB=clipself_amd/csrc/ab
(timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or folded or split_stream or fp8 or layernorm_forward_with" 2>&1 | tail -8) > $out/tests_ops.txt
tail -3 $out/tests_ops.txt
for r in 0 1; do
  for lib in $AB/libclipself_hip_cur.so clipself_amd/csrc/libclipself_hip.so $AB/libclipself_hip_prio.so $AB/libclipself_hip_touch.so; do
    GEMM_AB_NOREP=$( [ $r = 1 ] && echo 1 ) CLIPSELF_HIP_LIB=$lib timeout 300 python tools/gemm_ab.py 2048 1 "$(basename $lib)" 2>&1 | grep -v amdgpu.ids >> $out/gemm_ab.txt
  done
done
cat $out/gemm_ab.txt
for lib in $AB/libclipself_hip_cur.so clipself_amd/csrc/libclipself_hip.so $AB/libclipself_hip_cur.so clipself_amd/csrc/libclipself_hip.so; do
  echo "# $lib" >> $out/bench_ab.jsonl
  CLIPSELF_HIP_LIB=$lib timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{' >> $out/bench_ab.jsonl
done
python - $out/bench_ab.jsonl <<'PY'
import json, sys
lib = None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        lib = line[2:].strip()
    else:
        d = json.loads(line)
        print(f"{lib}: {d['value']:.1f} images/s, {d['ms_per_step']:.2f} ms/step, dominant kernel {d['roofline']['mean_us']:.0f} us")
PY
