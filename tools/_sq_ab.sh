cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/sq_mls.csv
export MLSGPU_SQ_KERNELS="processCornersCubeKernel<0, false>;"
bash tools/sq_counters_kernels.sh mls gpurun_out/sq_mls.csv
cat gpurun_out/sq_mls.csv | tail -24
