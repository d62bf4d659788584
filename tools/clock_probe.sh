#!/bin/bash
# Samples the GPU's clocks and power (rocm-smi, read-only) while bench.py runs: what does the chip sustain under the MFMA-bound loop?
cd "$(dirname "$0")/.."
python bench.py --steps 60 --warmup 3 --no-extras > gpurun_out/clock_probe_bench.log 2>&1 &
PID=$!
for i in $(seq 1 40); do
    kill -0 $PID 2>/dev/null || break
    echo "t=${i}s $(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e 's/.*level: //' -e 's/.*(W): /W /' | tr '\n' ' ')"
    sleep 1
done
wait $PID
tail -1 gpurun_out/clock_probe_bench.log | cut -c1-200
