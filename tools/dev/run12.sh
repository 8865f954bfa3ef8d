set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for cfg in magma_pythia_410M summit-1-3B 19M summit-70m-openclipH; do
for r in 1 2; do for v in 0 default; do
  if [ $v = 0 ]; then export MM_CU_SHARE=0; else unset MM_CU_SHARE; fi
  timeout -k 10 400 python -u bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/err.log | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$cfg cu_share=$v', d['ms_per_step'], d['value'], d['roofline']['frac'])" || { tail -5 gpurun_out/err.log; exit 1; }
done; done; done
