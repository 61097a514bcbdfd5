#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of processCorners for one MLS variant (no tracing with --pmc, see the prompt's rule).
# usage: bash tools/sq_counters.sh <variant> <label> <out.csv>
set -u
v=$1; label=$2; out=$3
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
args="bench.py --workload cfg2 --headline-only --no-timing --workers 1 --steps 1 --warmup 0 --variant $v"
rm -rf /tmp/sq_a_$v /tmp/sq_b_$v /tmp/sq_c_$v
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/sq_a_$v -o run -- python3 $args > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES -d /tmp/sq_b_$v -o run -- python3 $args > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS -d /tmp/sq_c_$v -o run -- python3 $args > /dev/null 2>&1
python3 tools/profile_summary.py sq "$label" "$out" /tmp/sq_a_$v /tmp/sq_b_$v /tmp/sq_c_$v
