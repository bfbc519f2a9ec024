#!/bin/bash
# sample power / clocks while the bench runs
(for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done) > gpurun_out/smi_samples.txt 2>&1 &
SMI=$!
python bench.py --steps 200 --warmup 5 --no-pmc --no-fp32-mode --no-parity --no-power > gpurun_out/smi_bench.txt 2>&1
wait $SMI
