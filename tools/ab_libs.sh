#!/bin/bash
# Developer A/B of two builds of libmm_hip.so on ONE box (boxes differ by up to 10 %): ab_tmp/libmm_hip_old.so is built from
# HEAD in a git worktree, ab_tmp/libmm_hip_new.so is the working tree's library.
#   here:        tools/ab_libs.sh prepare
#   on the box:  gpurun -- 'bash tools/ab_libs.sh run python -u tools/ab_attn.py'
set -e
if [ "$1" = prepare ]; then
  rm -rf /tmp/oldwt ab_tmp && mkdir -p ab_tmp
  git worktree add -q /tmp/oldwt HEAD
  make -C /tmp/oldwt/multimodal_amd/csrc -j8 > /dev/null
  cp /tmp/oldwt/multimodal_amd/libmm_hip.so ab_tmp/libmm_hip_old.so
  git worktree remove --force /tmp/oldwt
  make -C multimodal_amd/csrc -j8 > /dev/null
  cp multimodal_amd/libmm_hip.so ab_tmp/libmm_hip_new.so
  exit 0
fi
shift
for r in 1 2; do
  for v in old new; do
    cp ab_tmp/libmm_hip_$v.so multimodal_amd/libmm_hip.so
    echo "== $v (round $r)"
    timeout -k 10 200 "$@"
  done
done
cp ab_tmp/libmm_hip_new.so multimodal_amd/libmm_hip.so
