set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0; do
rm -rf /tmp/pk3
MM_BRANCH_STREAMS=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk3 -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/trace3.log 2>&1 || { tail -5 gpurun_out/trace3.log; exit 1; }
echo "== MM_BRANCH_STREAMS=$v"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/trace3.log
python3 tools/trace_steps.py /tmp/pk3 | tail -4
cp /tmp/pk3/*kernel_trace.csv gpurun_out/ktrace_$v.csv 2>/dev/null || find /tmp/pk3 -name "*kernel_trace.csv" -exec cp {} gpurun_out/ktrace_$v.csv \;
done
