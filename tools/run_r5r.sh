#!/bin/bash
# round 5, call R: go / no-go measurement for VERDICT r04 #4 (drop the render pass's re-encode): what the prune march costs when it also
# writes, for every encoded sample, the four per-encoding pairs of each level (16 B per level, level-major slab) and the composed
# 64-byte row (-DMARCH_STAGE=1, measurement-only build) against the shipped march, on one cached batch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5r
mkdir -p $OUT
cd $R
L=$OUT/log.txt
: > $L
export KB_WARM=${KB_WARM:-1500} KB_REPS=20 KB_CACHE=/tmp/kb_r5r.pt
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
for tag in default marchstage default marchstage; do
  lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
  echo "== lib=$tag mode=march" >> $L
  KB_LIB=$lib KB_ONLY=march timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$|march:" >> $L
done
echo "== lib=default mode=fwd" >> $L
KB_ONLY=fwd timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$" >> $L
