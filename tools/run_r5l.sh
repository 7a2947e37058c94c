#!/bin/bash
# round 5, call L: 10 000-step regime curve of the 1000-frame shape (configs[4]) with the bf16 MLP and the live replacer streaming from the
# pinned host capture; the whole GPU test suite on the final build.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5l
mkdir -p $OUT
cd $R
STEPS=10000 EVERY=2500 VIEWS=4 EXTRA="--frames 1000 --mlp-precision bf16 --host-capture-gb 24" timeout 900 python tools/long_run.py > $OUT/curve_frames1000.txt 2> $OUT/curve_frames1000.err
cat $OUT/curve_frames1000.txt | cut -c1-250
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log | cut -c1-250
