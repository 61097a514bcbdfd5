#!/bin/bash
# Runs ON THE GPU BOX (gpurun): where a wave of processCorners (variant 5) spends its cycles.  Build the library with
#   make -C mlsgpu_amd/csrc HIPFLAGS_EXTRA=-DMLSGPU_MLS5_CLOCK   (or compile mls.o by hand) and copy it to ab/clock.so, then
#   gpurun -- 'bash tools/mls_clock.sh [uniform|shells]'
# The plain kernel reads s_memtime at its barriers and leaves per-phase sums over all waves in the work-counter words 0-7
# (bench.py prints them with MLSGPU_BENCH_DUMP_MLS_COUNTERS=1): head, staging, barrier 1, compaction, tiles + drains, barrier 2,
# whole wave, number of waves.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp mlsgpu_amd/libmlsgpu_hip.so /tmp/orig.so
cp ab/${CLOCK_SO:-clock}.so mlsgpu_amd/libmlsgpu_hip.so
for cloud in "${@:-uniform}"; do
  MLSGPU_BENCH_DUMP_MLS_COUNTERS=1 timeout 300 python3 bench.py --headline-only --no-timing --no-cross-check --workers 1 --batch 4 --steps 1 --warmup 0 --dist $cloud 2> /tmp/clock_$cloud.err > /tmp/clock_$cloud.out
  grep "mls counters" /tmp/clock_$cloud.err | python3 -c "
import sys
w = [int(x) for x in sys.stdin.read().split()[2:10]]
names = ['head', 'staging', 'barrier 1', 'compaction', 'tiles + drains', 'barrier 2']
tot = w[6]
print('$cloud: %d waves, %.0f cycles per wave' % (w[7], tot / max(w[7], 1)))
for n, v in zip(names, w[:6]):
    print('  %-16s %6.1f %%  (%.0f cycles per wave)' % (n, 100.0 * v / tot, v / max(w[7], 1)))
print('  %-16s %6.1f %%' % ('rest', 100.0 * (tot - sum(w[:6])) / tot))
" | tee -a gpurun_out/mls_clock.txt
done
cp /tmp/orig.so mlsgpu_amd/libmlsgpu_hip.so
