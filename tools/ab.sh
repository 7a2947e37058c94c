#!/bin/bash
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-validation 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'enc/s', d['samples_encoded_by_prune_per_s'], 'frac', d['roofline']['frac'], 'S1', d['samples_per_ray_post'], 'S0', d['samples_per_ray_pre'], 'val', d['value'], 'psnr', d['train_psnr_db']); print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
