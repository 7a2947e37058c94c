#!/bin/bash
# round 5, call S: where to put the vector half of the backward relative to the binned table scatter: behind it, under emit + accumulate
# from the start (the engine's arrangement), or under the accumulate kernel alone (-DSB_MID_EVENT=1 measurement build).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5s
mkdir -p $OUT
cd $R
L=$OUT/log.txt
: > $L
export KB_WARM=${KB_WARM:-1500} KB_REPS=20 KB_CACHE=/tmp/kb_r5s.pt
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
for i in 1 2; do
  echo "== lib=midev mode=scattervec ($i)" >> $L
  KB_LIB=tools/_build/libhrf_hip_midev.so KB_ONLY=scattervec timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$|Error|error" >> $L
done
