set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 200 python -u tools/ab_splitk.py ab_tmp/libmm_hip_sk2048.so > $O/ab_splitk3.log 2>&1 || { tail -20 $O/ab_splitk3.log; exit 1; }
cat $O/ab_splitk3.log
bash tools/ab_libs.sh run python -u tools/ab_attn.py 2 > $O/ab_attn_accinit.log 2>&1 || { tail -20 $O/ab_attn_accinit.log; exit 1; }
grep -v amdgpu.ids $O/ab_attn_accinit.log
for r in 1 2; do for v in new notouch; do
  cp ab_tmp/libmm_hip_$v.so multimodal_amd/libmm_hip.so
  echo "== $v (round $r)"
  timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])" || exit 1
done; done > $O/ab_bench_touch.log 2>&1
cat $O/ab_bench_touch.log
cp ab_tmp/libmm_hip_new.so multimodal_amd/libmm_hip.so
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/pa1 -o a -- python3 /root/repo/tools/ab_attn.py 1 > /root/repo/$O/pmc_attn1.log 2>&1 || { tail -5 /root/repo/$O/pmc_attn1.log; exit 1; }
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM; do python3 /root/repo/tools/pmc_by_kernel.py /tmp/pa1 $c | grep -i "attn\|Kernel_Name"; done > /root/repo/$O/pmc_attn1.csv
cat /root/repo/$O/pmc_attn1.csv
