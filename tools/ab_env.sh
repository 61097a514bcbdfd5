#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the average duration of one kernel under several ENVIRONMENT settings of the same library,
# same box, same run:   gpurun -- 'bash tools/ab_env.sh "<kernel substring>" "<bench args>" base= m=MLSGPU_HIP_MLS_XCD_CHUNK=65535 ...'
# Each variant is name=VAR=value (or name= for the defaults).  See tools/ab_kernel.sh for the method.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
pat="$1"; shift
bargs="$1"; shift
for spec in "$@"; do
  name="${spec%%=*}"; setting="${spec#*=}"
  rm -rf /tmp/pp_$name
  if [ -n "$setting" ]; then export "$setting"; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp_$name -o run -- python3 bench.py --headline-only --no-timing --no-cross-check --workers 1 --batch 4 --steps 3 --warmup 1 $bargs > /dev/null 2>&1
  if [ -n "$setting" ]; then unset "${setting%%=*}"; fi
  python3 tools/profile_summary.py stats /tmp/pp_$name gpurun_out/pp_$name.csv "$name" > /dev/null
  echo "$name: $(grep -i "$pat" gpurun_out/pp_$name.csv | head -3 | tr '\n' ' ')"
done
