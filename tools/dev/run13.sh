set -o pipefail
cp ab_tmp/libmm_hip_tn4.so multimodal_amd/libmm_hip.so
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 120 --timeout-method thread -k "gemm_tn or adapter or tn" 2>&1 | tail -2 || exit 1
timeout -k 10 120 python -u tools/ab_tn_pair.py ab_tmp/libmm_hip_pol.so ab_tmp/libmm_hip_tn4.so 2>&1 | grep -v amdgpu.ids
