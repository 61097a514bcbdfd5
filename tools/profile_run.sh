#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the three rocprofv3 passes DESIGN.md section 7 cites, summarised into gpurun_out/profiles_<tag>/
# (the raw .db files stay in /tmp: they are tens of MB each).  usage: bash tools/profile_run.sh r02
set -u
tag=${1:-rNN}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_$tag
mkdir -p "$out"
cmd="bench.py --headline-only --no-timing --workers 1 --steps 5 --warmup 1"
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o run -- python3 $cmd > "$out/stats_run.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_fetch_$tag -o run -- python3 bench.py --headline-only --no-timing --workers 1 --steps 1 --warmup 0 > "$out/fetch_run.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_write_$tag -o run -- python3 bench.py --headline-only --no-timing --workers 1 --steps 1 --warmup 0 > "$out/write_run.log" 2>&1
python3 tools/profile_summary.py stats /tmp/prof_$tag "$out/${tag}_cfg3_kernel_stats.csv" "python3 $cmd"
python3 tools/profile_summary.py traffic /tmp/pmc_fetch_$tag /tmp/pmc_write_$tag "$out/${tag}_cfg3_pmc_hbm_traffic.csv"
cp profiles/traffic.json "$out/traffic.json"
tail -2 "$out/stats_run.log"
