mkdir -p gpurun_out/r03e
rm -f gpurun_out/parity_metrics.txt gpurun_out/step_metrics.txt
(timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_fullsize.py -m gpu -x -q -k "layernorm or colsum or swiglu_cast or step or unlocked or two_rank or reproducible or cfg1 or cfg4 or openai or regionclip or loss_curve or removed_schedules" 2>&1 | tail -15) > gpurun_out/r03e/tests.txt
cp gpurun_out/step_metrics.txt gpurun_out/r03e/ 2>/dev/null; cp gpurun_out/parity_metrics.txt gpurun_out/r03e/ 2>/dev/null
python tools/step_phases.py > gpurun_out/r03e/phases.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03e/bench.json 2> gpurun_out/r03e/bench.err
tail -6 gpurun_out/r03e/tests.txt; tail -1 gpurun_out/r03e/phases.txt; cut -c1-300 gpurun_out/r03e/bench.json
