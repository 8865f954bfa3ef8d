export TMPDIR=/tmp
cd /tmp
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVE_CYCLES"
timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pl -o a -- python3 /root/repo/tools/bench_kernels.py attn gemm > /root/repo/gpurun_out/pmc_lds.log 2>&1 || { tail -5 /root/repo/gpurun_out/pmc_lds.log; exit 1; }
for c in $C; do python3 /root/repo/tools/pmc_by_kernel.py /tmp/pl $c | grep -i "attn\|gemm_nt\|Kernel_Name" | cut -c1-60,200-; done > /root/repo/gpurun_out/pmc_lds.csv
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob('/tmp/pl/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:70]
        a=acc[k][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,d in acc.items():
    if 'attn' in k or 'gemm_nt' in k:
        print(k)
        for c,(n,s) in d.items(): print("   %-28s %14.0f per launch (%d)"%(c,s/n,n))
PY
