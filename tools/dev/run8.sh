set -o pipefail
python - <<'PY'
s=open('tools/decompose_gemm256.py').read()
s=s.replace("for dbg in (0, 4, 5, 6, 7, 16, 20, 17, 18, 21, 22, 23, 16, 0):","for dbg in (16, 20, 17, 21, 23, 16, 20):")
open('/tmp/dec.py','w').write(s.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))","'/root/repo'"))
PY
timeout -k 10 120 python -u /tmp/dec.py 8192 50432 1024 0 2>&1 | grep -v amdgpu.ids &&
timeout -k 10 120 python -u /tmp/dec.py 8192 16384 1024 0 2>&1 | grep -v amdgpu.ids &&
timeout -k 10 120 python -u /tmp/dec.py 8192 16384 4096 0 2>&1 | grep -v amdgpu.ids
