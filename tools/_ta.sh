cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/ta_tri.csv
export MLSGPU_SQ_KERNELS="latticeTrianglesRow;"
args="bench.py --headline-only --no-timing --no-cross-check --workers 1 --steps 1 --warmup 0"
export MLSGPU_HIP_TRI_PIPE=1
rm -rf /tmp/ta_a /tmp/ta_b /tmp/ta_c
date +%s
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -d /tmp/ta_a -o run -- python3 $args > /dev/null 2>&1; echo rc $?; date +%s
timeout 120 rocprofv3 --pmc TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum -d /tmp/ta_b -o run -- python3 $args > /dev/null 2>&1; echo rc $?; date +%s
timeout 120 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum -d /tmp/ta_c -o run -- python3 $args > /dev/null 2>&1; echo rc $?; date +%s
python3 tools/profile_summary.py sq pipe gpurun_out/ta_tri.csv /tmp/ta_a /tmp/ta_b /tmp/ta_c
cat gpurun_out/ta_tri.csv
