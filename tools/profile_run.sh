#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the rocprofv3 passes DESIGN.md section 7 cites (one worker, FOUR buckets per octree launch: the
# shape every round's per-launch figures are quoted on, whatever bench.py's default batch is), summarised into gpurun_out/profiles_<tag>/
# (the raw .db files stay in /tmp: they are tens of MB each).  usage: bash tools/profile_run.sh r03
# One pass per counter set, never --pmc together with a trace (the pool's rule); python3 itself after `--`.
set -u
tag=${1:-rNN}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_$tag
mkdir -p "$out"
cmd="bench.py --headline-only --no-timing --no-cross-check --workers 1 --batch 4 --steps 5 --warmup 1"
one="bench.py --headline-only --no-timing --no-cross-check --workers 1 --batch 4 --steps 1 --warmup 0"
rm -rf /tmp/prof_$tag /tmp/pmc_rd_$tag /tmp/pmc_wr_$tag /tmp/pmc_fetch_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o run -- python3 $cmd > "$out/stats_run.log" 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/pmc_rd_$tag -o run -- python3 $one > "$out/rd_run.log" 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d /tmp/pmc_wr_$tag -o run -- python3 $one > "$out/wr_run.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_fetch_$tag -o run -- python3 $one > "$out/fetch_run.log" 2>&1
python3 tools/profile_summary.py stats /tmp/prof_$tag "$out/${tag}_cfg3_kernel_stats.csv" "python3 $cmd"
python3 tools/profile_summary.py traffic_resolved /tmp/pmc_rd_$tag /tmp/pmc_wr_$tag "$out/${tag}_cfg3_pmc_hbm_traffic.csv"
cp profiles/traffic.json "$out/traffic.json"
python3 - <<P
import sys
sys.path.insert(0, "tools")
import profile_summary as ps
f, c = ps.counters("/tmp/pmc_fetch_$tag", "FETCH_SIZE")
open("$out/${tag}_cfg3_fetch_size_kb.csv", "w").write("# rocprofv3 --pmc FETCH_SIZE (KiB, as reported, uncorrected), same command\nkernel,calls,FETCH_SIZE_KB_total\n" + "".join('"%s",%d,%.1f\n' % (k, c[k], f[k]) for k in sorted(f, key=lambda k: -f[k])))
P
tail -2 "$out/stats_run.log"
