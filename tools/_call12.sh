out=gpurun_out/r03l
mkdir -p $out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $out/tests.txt
tail -5 $out/tests.txt
timeout 600 python tools/regionclip_bench.py 5 > $out/regionclip.jsonl 2> $out/regionclip.err
cat $out/regionclip.jsonl | cut -c1-400
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
cut -c1-600 $out/bench.json
