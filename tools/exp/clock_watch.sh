#!/bin/bash
# samples sclk / power with rocm-smi while bench.py runs (is the fused block power-throttled?)
python bench.py --steps 1500 --warmup 5 --no-cpu-baseline > gpurun_out/clock_bench.json 2>/dev/null &
pid=$!
while kill -0 $pid 2>/dev/null; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr '\n' ' '
    echo
    sleep 0.25
done | sort | uniq -c | sort -rn | head -12
cut -c1-120 gpurun_out/clock_bench.json
