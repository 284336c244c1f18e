#!/bin/bash
# B = 8 / 32 / 128 headline-style runs of bench.py (no extra legs, no CPU baseline): value and ms per step
# usage: scripts/small_batch_ab.sh <tag> [batches...]
tag=$1; shift
for b in "${@:-8 32}"; do
  python bench.py --batch $b --steps 5 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag B=$b', d['value'], d['ms_per_step'], d.get('value_no_overlap'))"
done
