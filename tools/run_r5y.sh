#!/bin/bash
# round 5, call Y: the emit kernel's walk as one straight-line body per step (-DSB_FLAT_WALK=1: 204 instead of 289 vector instructions per
# step on hashed levels, 300 instead of 466 on dense ones) against the shipped branching walk: binned scatter on one cached batch, then
# the scatter tests on the variant library.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5y
mkdir -p $OUT
cd $R
L=$OUT/log.txt
: > $L
export KB_WARM=${KB_WARM:-1500} KB_REPS=20 KB_CACHE=/tmp/kb_r5y.pt
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
for tag in default flatwalk default flatwalk; do
  lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
  echo "== lib=$tag mode=scattervec" >> $L
  KB_LIB=$lib KB_ONLY=scattervec timeout 120 python tools/kbench.py 2>&1 | grep -E "tables alone|under emit" >> $L
done
HRF_TEST_LIB=tools/_build/libhrf_hip_flatwalk.so timeout 400 python -m pytest tests/test_gpu_scatter.py -x -q -m gpu >> $L 2>&1
echo "pytest flatwalk rc=$?" >> $L
