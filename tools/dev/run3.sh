set -o pipefail
for v in $VARIANTS; do
  cp ab_tmp/libmm_hip_$v.so multimodal_amd/libmm_hip.so
  echo "== $v"
  timeout -k 10 300 python -u -m pytest tests/test_attention_gpu.py tests/test_attention_dropout_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -1
  timeout -k 10 200 python -u tools/ab_attn.py 3 2>&1 | grep -v amdgpu.ids
done
