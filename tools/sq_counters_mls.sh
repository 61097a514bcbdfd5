#!/bin/bash
# Runs ON THE GPU BOX: SQ counters (vector, LDS, matrix pipe, waits) of the TIMED processCorners kernel of one MLS variant over
# one pass of cfg3 (no tracing with --pmc).   usage: bash tools/sq_counters_mls.sh <variant> <label> <out.csv> [bench args]
set -u
v=$1; label=$2; out=$3; shift 3
case $v in 5) k="processCornersMatrixKernel<0, false>";; 4) k="processCornersCubeKernel<0, false>";; *) k="processCornersKernel<0, false>";; esac
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
args="bench.py --headline-only --no-timing --no-cross-check --workers 1 --batch 4 --steps 1 --warmup 0 --variant $v $*"
rm -rf /tmp/sqm_a /tmp/sqm_b /tmp/sqm_c /tmp/sqm_d
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/sqm_a -o run -- python3 $args > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES -d /tmp/sqm_b -o run -- python3 $args > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 -d /tmp/sqm_c -o run -- python3 $args > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_LEVEL_WAVES SQ_INSTS_BRANCH -d /tmp/sqm_d -o run -- python3 $args > /dev/null 2>&1
MLSGPU_SQ_KERNELS="$k;" MLSGPU_SQ_JSON="${MLSGPU_SQ_JSON:-gpurun_out/sq.json:cfg3/uniform/processCorners/variant$v}" python3 tools/profile_summary.py sq "$label" "$out" /tmp/sqm_a /tmp/sqm_b /tmp/sqm_c /tmp/sqm_d
