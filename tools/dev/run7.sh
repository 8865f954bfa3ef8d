set -o pipefail
A=ab_tmp/libmm_hip_${BASE:-cur}.so; B=ab_tmp/libmm_hip_${VAR:-rf}.so
cp $B multimodal_amd/libmm_hip.so
timeout -k 10 400 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 120 --timeout-method thread -k "gemm" 2>&1 | tail -2 || exit 1
cp $A multimodal_amd/libmm_hip.so
for shp in "8192 4096 1024 2" "8192 4096 1024 6 1" "8192 3072 1024 1" "8192 1024 4096 0" "8192 1024 1024 0" "8192 2048 16384 0" "8192 50304 1024 0"; do
timeout -k 10 120 python -u tools/ab_two_libs.py $A $B $shp 2>&1 | grep -v amdgpu.ids || exit 1
done
