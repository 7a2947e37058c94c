#!/bin/bash
for v in simple paired; do
  if [ $v = simple ]; then export HRF_GATHER_SIMPLE=1; else unset HRF_GATHER_SIMPLE; fi
  python bench.py --no-cpu-baseline --no-validation --pretrain 1500 --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'ms/step', d['ms_per_step'], 'march ms/step', d['kernel_ms_per_step'], 'enc/s', d['samples_encoded_by_prune_per_s'], 'frac', d['roofline']['frac'], 'S1', d['samples_per_ray_post'])"
done
