#!/bin/bash
# Runs ON THE GPU BOX: the headline (ms per step of the timed region) under several worker / batch / group settings, two rounds.
#   gpurun -- 'bash tools/sweep_settings.sh "--workers 2 --batch 6 --marching-group 2" "--workers 3 --batch 6 --marching-group 2" ...'
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for s in "$@"; do
    python3 bench.py --headline-only --no-timing --no-cross-check --steps 60 --warmup 5 $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s', d['value'], d['ms_per_step'])"
  done
done
