cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
wc -l gpurun_out/counters_list.txt
grep -o "TA_[A-Z_0-9a-z]*\|TCP_[A-Z_0-9a-z]*\|TD_[A-Z_0-9a-z]*" gpurun_out/counters_list.txt | sort -u | tr '\n' ' ' | head -c 6000
