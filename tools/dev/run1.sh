set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_ops_gpu.py tests/test_attention_gpu.py tests/test_attention_dropout_gpu.py -k "splitk or attention or attn" -x -q --timeout 120 --timeout-method thread > $O/r3_t3.log 2>&1 || { tail -30 $O/r3_t3.log; exit 1; }
tail -3 $O/r3_t3.log
timeout -k 10 200 python -u tools/ab_two_libs.py multimodal_amd/libmm_hip.so ab_tmp/libmm_hip_notouch.so 8192 4096 1024 6 1 > $O/ab_touch.log 2>&1 || { tail -20 $O/ab_touch.log; exit 1; }
cat $O/ab_touch.log
timeout -k 10 200 python -u tools/ab_splitk.py > $O/ab_splitk2.log 2>&1 || { tail -20 $O/ab_splitk2.log; exit 1; }
cat $O/ab_splitk2.log
bash tools/ab_libs.sh run python -u tools/ab_attn.py 2 > $O/ab_attn_accinit.log 2>&1 || { tail -20 $O/ab_attn_accinit.log; exit 1; }
cat $O/ab_attn_accinit.log
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/pa1 -o a -- python3 /root/repo/tools/ab_attn.py 1 > /root/repo/$O/pmc_attn1.log 2>&1 || { tail -5 /root/repo/$O/pmc_attn1.log; exit 1; }
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM; do python3 /root/repo/tools/pmc_by_kernel.py /tmp/pa1 $c | grep -i "attn\|Kernel_Name"; done > /root/repo/$O/pmc_attn1.csv
cat /root/repo/$O/pmc_attn1.csv
