set -o pipefail
A=ab_tmp/libmm_hip_${BASE:-old}.so; B=ab_tmp/libmm_hip_${VAR:-pol}.so
for shp in "8192 4096 1024 2" "8192 4096 1024 6 1" "8192 3072 1024 1" "8192 1024 4096 0" "8192 1024 1024 0" "8192 50304 1024 0"; do
timeout -k 10 120 python -u tools/ab_two_libs.py $A $B $shp 2>&1 | grep -v amdgpu.ids || exit 1
done
