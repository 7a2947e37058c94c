#!/bin/bash
# round 5, call P: the fused MLP backward with two 16-sample tiles in flight per wavefront (-DMLPB_PAIR=1) against the shipped kernel:
# time on one cached batch, then the MLP / engine parity tests on the variant library.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5p
mkdir -p $OUT
cd $R
L=$OUT/log.txt
: > $L
export KB_WARM=${KB_WARM:-1500} KB_REPS=20 KB_CACHE=/tmp/kb_r5p.pt
KB_ONLY=none timeout 300 python tools/kbench.py > $OUT/kb_warm.log 2>&1
for tag in default mlppair default mlppair; do
  lib=tools/_build/libhrf_hip_$tag.so; [ $tag = default ] && lib=""
  echo "== lib=$tag mode=mlpbwd" >> $L
  KB_LIB=$lib KB_ONLY=mlpbwd timeout 120 python tools/kbench.py 2>&1 | grep -E "ms$" >> $L
done
HRF_TEST_LIB=tools/_build/libhrf_hip_mlppair.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_gpu_round4.py -x -q -m gpu -k "mlp or bwd or backward or engine or knob or reference or differentiable or train" >> $L 2>&1
echo "pytest mlppair rc=$?" >> $L
