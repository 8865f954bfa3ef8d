#!/bin/bash
# Developer A/B of named library builds on ONE box: tools/ab_libs2.sh "new hoist" python -u tools/ab_attn.py 2
# (ab_tmp/libmm_hip_<name>.so each; built here with make ... OUT=ab_tmp/libmm_hip_<name>.so EXTRA=-DMM_AB_...)
set -e
NAMES=$1; shift
for r in 1 2; do
  for v in $NAMES; do
    cp ab_tmp/libmm_hip_$v.so multimodal_amd/libmm_hip.so
    echo "== $v (round $r)"
    timeout -k 10 300 "$@"
  done
done
cp ab_tmp/libmm_hip_new.so multimodal_amd/libmm_hip.so
