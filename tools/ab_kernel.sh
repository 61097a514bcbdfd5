#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the average duration of one kernel under several builds of the library, same box, same run.
# Build each variant here, copy it to ab/<name>.so (ab/ is git-ignored but travels with gpurun), then
#   gpurun -- 'bash tools/ab_kernel.sh "<kernel name substring>" base variant1 variant2 base'
# rocprofv3 --kernel-trace --stats over three passes of cfg3 on one worker: run-to-run spread of a kernel's average is ~0.05 %
# on one box (boxes differ by 1-3 %), so a 0.5 % change is visible.  The bench refuses to report a wrong digest AFTER the
# kernels ran, so timing-only builds (a store dropped, a phase skipped) can be measured as well.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
pat="$1"; shift
cp mlsgpu_amd/libmlsgpu_hip.so /tmp/orig.so
for v in "$@"; do
  cp ab/$v.so mlsgpu_amd/libmlsgpu_hip.so
  rm -rf /tmp/pp_$v
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pp_$v -o run -- python3 bench.py --headline-only --no-timing --no-cross-check --workers 1 --batch 4 --steps 3 --warmup 1 > /dev/null 2>&1
  python3 tools/profile_summary.py stats /tmp/pp_$v gpurun_out/pp_$v.csv "$v" > /dev/null
  echo "$v: $(grep -i "$pat" gpurun_out/pp_$v.csv | head -3 | tr '\n' ' ')"
done
cp /tmp/orig.so mlsgpu_amd/libmlsgpu_hip.so
