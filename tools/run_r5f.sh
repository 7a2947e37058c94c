#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5f
mkdir -p $OUT
cd $R
free -g | head -2 > $OUT/log.txt
python tools/dbg_r5.py >> $OUT/log.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py -q -m gpu -x >> $OUT/log.txt 2>&1
tail -60 $OUT/log.txt | cut -c1-250 | grep -v amdgpu.ids
