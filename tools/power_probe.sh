#!/bin/bash
# Developer probe: sample clocks / power (read-only rocm-smi queries) while bench.py replays the captured step.
python -u bench.py --steps 400 --warmup 3 --no-cpu-baseline > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 11
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "sclk|Socket|Max Graphics|Average" | sed 's/GPU\[0\]//;s/\t//g' | tr '\n' '|'
  echo
  sleep 0.8
done
wait $BP
tail -1 gpurun_out/power_bench.log | cut -c1-200
