#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the kernels named in MLSGPU_SQ_KERNELS (comma-separated substrings) over one pass of a
# workload (no tracing with --pmc).  usage: MLSGPU_SQ_KERNELS=a,b bash tools/sq_counters_kernels.sh <label> <out.csv> <bench args...>
set -u
label=$1; out=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
args="bench.py --headline-only --no-timing --no-cross-check --workers 1 --batch 4 --steps 1 --warmup 0 $*"
rm -rf /tmp/sqk_a /tmp/sqk_b /tmp/sqk_c
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/sqk_a -o run -- python3 $args > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES -d /tmp/sqk_b -o run -- python3 $args > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM -d /tmp/sqk_c -o run -- python3 $args > /dev/null 2>&1
python3 tools/profile_summary.py sq "$label" "$out" /tmp/sqk_a /tmp/sqk_b /tmp/sqk_c
