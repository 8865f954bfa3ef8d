for v in $VARIANTS; do
  cp ab_tmp/libmm_hip_$v.so multimodal_amd/libmm_hip.so
  echo "== $v"
  timeout -k 10 200 python -u tools/bench_hbm_kernels.py $ARGS 2>&1 | grep -v amdgpu.ids
done
cp ab_tmp/libmm_hip_ln4.so multimodal_amd/libmm_hip.so
