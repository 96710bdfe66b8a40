out=gpurun_out/r03s
mkdir -p $out
rm -f gpurun_out/parity_metrics.txt gpurun_out/step_metrics.txt gpurun_out/ops_metrics.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $out/tests.txt
tail -3 $out/tests.txt
cp gpurun_out/parity_metrics.txt gpurun_out/step_metrics.txt gpurun_out/ops_metrics.txt $out/ 2>/dev/null
python tools/step_phases.py > $out/phases.txt 2>&1
tail -2 $out/phases.txt
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
cut -c1-300 $out/bench.json
bash tools/profile_round.sh r03s_prof > $out/profile_round.log 2>&1
ls gpurun_out/r03s_prof
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench2.json 2> $out/bench2.err
cut -c1-200 $out/bench2.json
