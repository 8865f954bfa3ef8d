#!/bin/bash
# The round's profile set on the GPU box (run from the repo root): bash tools/profile_round.sh r01_k
# Kernel trace of the bench, three separate PMC passes (FETCH_SIZE / WRITE_SIZE / matrix-pipe counters), one unprofiled line.
# The profiled passes keep the layer on ONE stream (MM_BRANCH_STREAMS=0): with two, kernels of the two branches share the CUs and
# a kernel's traced duration is no longer the kernel's own time.  The unprofiled line runs the default (two streams).
set -e
TAG=$1
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 python3 /root/repo/bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench_unprofiled.log 2>&1
grep '^{' $OUT/${TAG}_bench_unprofiled.log | tail -1 > $OUT/${TAG}_bench_unprofiled_line.json
MM_BRANCH_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o bench -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_trace.log 2>&1
cp /tmp/pk/bench_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
# the profiler prints after the program: the bench line is the last line that STARTS with '{', not the log's last line
(grep '^{' $OUT/${TAG}_trace.log | tail -1 > $OUT/${TAG}_bench_line.json) || true
# the same trace with the layer's two branch streams (the form the timed region replays: kernels of the two branches overlap, and
# inside the region each branch's 256-wide GEMMs run with half the CUs - durations here are NOT a kernel's own time)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk2 -o bench -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_trace_two_streams.log 2>&1
cp /tmp/pk2/bench_kernel_stats.csv $OUT/${TAG}_two_stream_kernel_stats.csv
MM_BRANCH_STREAMS=0 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/${TAG}_pmc_fetch.log 2>&1
python3 /root/repo/tools/pmc_by_kernel.py /tmp/pf FETCH_SIZE > $OUT/${TAG}_pmc_fetch_by_kernel.csv
MM_BRANCH_STREAMS=0 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/${TAG}_pmc_write.log 2>&1
python3 /root/repo/tools/pmc_by_kernel.py /tmp/pw WRITE_SIZE > $OUT/${TAG}_pmc_write_by_kernel.csv
MM_BRANCH_STREAMS=0 timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -o m -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/${TAG}_pmc_mfma.log 2>&1
python3 /root/repo/tools/pmc_mfma_table.py /tmp/pm > $OUT/${TAG}_pmc_mfma.md
echo done
