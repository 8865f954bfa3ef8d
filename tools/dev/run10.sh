set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $VARIANTS; do
  cp ab_tmp/libmm_hip_$v.so multimodal_amd/libmm_hip.so
  rm -rf gpurun_out/prof_$v
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$v -o p -- python bench.py --steps 9 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$v.log 2>&1 || exit 1
  find gpurun_out/prof_$v -name "*kernel_stats.csv" | head -1
done
