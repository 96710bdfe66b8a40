mkdir -p gpurun_out/r03d
rm -f gpurun_out/parity_metrics.txt gpurun_out/step_metrics.txt gpurun_out/ops_metrics.txt
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r03d/tests.txt
for f in parity_metrics step_metrics ops_metrics; do cp gpurun_out/$f.txt gpurun_out/r03d/$f.txt 2>/dev/null; done
python bench.py --steps 20 --warmup 5 > gpurun_out/r03d/bench.json 2> gpurun_out/r03d/bench.err
bash tools/profile_round.sh r03d > gpurun_out/r03d/profile.log 2>&1
tail -5 gpurun_out/r03d/tests.txt; cat gpurun_out/r03d/bench.json
