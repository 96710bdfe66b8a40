out=gpurun_out/r03z
mkdir -p $out
rm -f gpurun_out/parity_metrics.txt gpurun_out/step_metrics.txt gpurun_out/ops_metrics.txt
(timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $out/tests.txt
tail -3 $out/tests.txt
cp gpurun_out/parity_metrics.txt gpurun_out/step_metrics.txt $out/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python tools/step_phases.py > $out/phases.txt 2>&1; tail -1 $out/phases.txt
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; cut -c1-260 $out/bench.json
bash tools/profile_round.sh r03z_prof > $out/profile_round.log 2>&1
ls gpurun_out/r03z_prof | tr '\n' ' '
