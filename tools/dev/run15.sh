set -o pipefail
for cfg in 19M summit-70m-openclipH magma_pythia_410M summit-1-3B 6-7B; do
  timeout -k 10 500 python -u bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/err.log | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'], d['mfma_util_model'], d['roofline']['kernel'], d['roofline']['frac'])" || { tail -5 gpurun_out/err.log; }
done
