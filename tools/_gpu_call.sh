cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
(timeout 600 python -m pytest tests/test_gpu_tree.py tests/test_gpu_bucket.py -x -q -m gpu) 2>&1 | tail -2
rm -rf /tmp/prof_f
rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o run -- python3 bench.py --headline-only --no-timing --workers 1 --steps 5 --warmup 1 > gpurun_out/r03/prof_f.log 2>&1
python3 tools/profile_summary.py stats /tmp/prof_f gpurun_out/r03/fused_kernel_stats.csv "fused"
grep -E "entry|sortDigit|sortHist|sortScatter" gpurun_out/r03/fused_kernel_stats.csv | cut -c1-150
for f in 0 1; do
MLSGPU_HIP_OCTREE_FUSED=$f python bench.py --headline-only --steps 60 > gpurun_out/r03/head_fused$f.json 2> gpurun_out/r03/head_fused$f.err
python - <<P
import json
d=json.loads(open('gpurun_out/r03/head_fused$f.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('fused $f', d['value'], d['ms_per_step'], 'compute', k['device.compute'], 'entries', k['kernel.octree.writeEntries.time'], 'sort', k['kernel.octree.sort.time'], d['output_digest'].get('ok'))
P
done
