set -o pipefail
for r in 1 2; do for v in $VARIANTS; do
  cp ab_tmp/libmm_hip_$v.so multimodal_amd/libmm_hip.so
  timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline_by_time'); print('$v', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_kernel_ms'], r['achieved'] if r else None)" || exit 1
done; done
cp ab_tmp/libmm_hip_cur.so multimodal_amd/libmm_hip.so
