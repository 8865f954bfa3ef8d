set -o pipefail
for r in 1 2; do for v in 0 1; do
  MM_WGRAD_STREAM=$v timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/err_$v.log | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('wgrad_stream=$v', d['ms_per_step'], d['value'])" || { tail -5 gpurun_out/err_$v.log; exit 1; }
done; done
