#!/bin/bash
# Developer probe: sample clocks / power (read-only rocm-smi queries) while bench.py replays the captured step.
python -u bench.py --steps 300 --warmup 3 --no-cpu-baseline > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power|mclk" | head -4 | tr '\n' ' '
  echo
  sleep 0.7
done
wait $BP
tail -1 gpurun_out/power_bench.log | cut -c1-200
