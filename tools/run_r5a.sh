#!/bin/bash
# round 5, call A: the rewritten level body (enc_level_shared) -- bit-equality tests, then march / render-pass encode timings
# of the new library against the round-4 one (tools/_build/libhrf_hip_base.so) on one cached batch.
mkdir -p gpurun_out
L=gpurun_out/r5a.log
: > $L
export KB_CACHE=/tmp/kb.pt KB_WARM=${KB_WARM:-1500} KB_REPS=20
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shared_level_body or encode4d_forward or fused_prune_march or segment_schedule" >> $L 2>&1
echo "pytest rc=$?" >> $L
for only in fwd march scatterprof; do
  for lib in "" tools/_build/libhrf_hip_base.so; do
    echo "== KB_ONLY=$only KB_LIB=$lib" >> $L
    KB_ONLY=$only KB_LIB=$lib python tools/kbench.py 2>&1 | grep -v "^$" | tail -8 >> $L
  done
done
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_ref_fixtures.py -x -q -m gpu >> $L 2>&1
echo "pytest2 rc=$?" >> $L
tail -60 $L
