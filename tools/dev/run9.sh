set -o pipefail
A=ab_tmp/libmm_hip_ring3.so
for v in T1 T2; do
B=ab_tmp/libmm_hip_$v.so
timeout -k 10 120 python -u tools/ab_two_libs.py $A $B 8192 4096 1024 2 2>&1 | grep -v amdgpu.ids &&
timeout -k 10 120 python -u tools/ab_two_libs.py $A $B 8192 2048 16384 0 2>&1 | grep -v amdgpu.ids || exit 1
done
python - <<PY
s=open("tools/decompose_gemm256.py").read()
s=s.replace("for dbg in (0, 4, 5, 6, 7, 16, 20, 17, 18, 21, 22, 23, 16, 0):","for dbg in (16, 20, 17, 18, 21, 22, 23, 16):")
open("/tmp/dec.py","w").write(s.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))","\"/root/repo\""))
PY
timeout -k 10 120 python -u /tmp/dec.py 8192 2048 16384 0 2>&1 | grep -v amdgpu.ids
