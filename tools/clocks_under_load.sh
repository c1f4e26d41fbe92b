#!/bin/bash
# What the GPU's clocks and power do WHILE the bench runs (boxes of this pool run the same binary 25 % apart)
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | tr '\n' ';'; echo; sleep 0.5; done ) > gpurun_out/r2_clocks.txt &
SM=$!
python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-transfers > gpurun_out/r2_clocks_bench.json 2>/dev/null
wait $SM
python3 -c "import json; d=json.loads(open('gpurun_out/r2_clocks_bench.json').read()); print('bench', d['value'], 'Mpix/s; placement', d['placement_trials_ms'], 'avg launch', d['roofline']['avg_launch_ms'])"
sort gpurun_out/r2_clocks.txt | uniq -c | sort -rn | head -8
