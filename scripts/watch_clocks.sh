#!/bin/bash
# Samples GPU clock / power with rocm-smi while the bench workload runs (sustained fp32-MFMA load): usage scripts/watch_clocks.sh out_dir
out=${1:-gpurun_out/clocks}; mkdir -p "$out"
rocm-smi --showclocks --showpower --showtemp > "$out/idle.txt" 2>&1
python bench.py --steps 6 --warmup 0 --prewarm_s 30 --no_cpu_baseline --side_steps 0 > "$out/bench.json" 2> "$out/bench.err" &
pid=$!
for i in $(seq 1 14); do sleep 5; echo "--- t=$((i*5))s" >> "$out/load.txt"; rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" >> "$out/load.txt"; done
wait $pid
cat "$out/load.txt" | grep -E "t=|sclk|Power" | head -60
python -c "import json; d=json.load(open('$out/bench.json')); print('value', d['value'], 'job', d['roofline']['job_gemm_tflops'])"
