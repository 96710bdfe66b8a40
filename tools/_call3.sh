mkdir -p gpurun_out/r03c
rm -f gpurun_out/parity_metrics.txt gpurun_out/step_metrics.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py tests/test_gpu_step.py -m gpu -x -q -k "parity or kernel_rounding or teacher_forced or frozen or roialign or rccl or two_rank or end_to_end or tiny_tower" 2>&1 | tail -25) > gpurun_out/r03c/tests.txt
cp gpurun_out/parity_metrics.txt gpurun_out/r03c/parity_metrics.txt 2>/dev/null
cp gpurun_out/step_metrics.txt gpurun_out/r03c/step_metrics.txt 2>/dev/null
cat gpurun_out/r03c/tests.txt; cat gpurun_out/r03c/parity_metrics.txt | cut -c1-400
