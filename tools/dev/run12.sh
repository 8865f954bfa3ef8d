set -o pipefail
for r in 1 2; do for v in 0,128 0,64 0,96 0,160 0,176 0,192 0,0; do
  MM_CU_SHARE=$v timeout -k 10 400 python -u bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>gpurun_out/err.log | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('cu_share=$v', d['ms_per_step'], d['value'])" || { tail -5 gpurun_out/err.log; exit 1; }
done; done
