#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the average duration of processCorners under several MLS kernel variants of ONE library,
# same box, same run:   gpurun -- 'bash tools/ab_variant.sh 4 5 4 5'
# (tools/ab_kernel.sh compares builds of the library; this compares the kernels one build holds.)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in "$@"; do
  rm -rf /tmp/pv_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv_$v -o run -- python3 bench.py --headline-only --no-timing --no-cross-check --workers 1 --batch 4 --steps 3 --warmup 1 --variant $v ${AB_EXTRA} > /tmp/pv_$v.log 2>&1
  python3 tools/profile_summary.py stats /tmp/pv_$v gpurun_out/pv_$v.csv "variant $v" > /dev/null
  echo "variant $v: $(grep -i "processCorners" gpurun_out/pv_$v.csv | head -3 | tr '\n' ' ')"
  tail -2 /tmp/pv_$v.log | cut -c1-600
done
