#!/bin/bash
# Round 5: row-major V in the short-sequence attention forward -- tests, micro-benchmark (tools/attn_bench.py, interleaved with the old form), step A/B.
tag=${1:-r05h}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" > "$out/tests.log" 2>&1; echo "attention tests rc $?" >> "$out/tests.log"
for r in 1 2; do
  echo "# transposed V image" >> "$out/attn_bench.txt"; CS_ATTN_FWD8_VT=1 timeout 300 python tools/attn_bench.py 2048 2>&1 | grep attn_fwd >> "$out/attn_bench.txt"
  echo "# row-major V" >> "$out/attn_bench.txt"; timeout 300 python tools/attn_bench.py 2048 2>&1 | grep attn_fwd >> "$out/attn_bench.txt"
done
for r in 1 2; do
  CS_ATTN_FWD8_VT=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('V^T image rep $r: %.1f images/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> "$out/bench_ab.txt"
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('row-major V rep $r: %.1f images/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> "$out/bench_ab.txt"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x >> "$out/tests.log" 2>&1; echo "parity rc $?" >> "$out/tests.log"
tail -6 "$out/tests.log"; cat "$out/attn_bench.txt" "$out/bench_ab.txt"
