cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_marching.py tests/test_gpu_bucket.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
  rm -rf /tmp/pm_$rep
  rocprofv3 --kernel-trace --stats -d /tmp/pm_$rep -o run -- python3 bench.py --headline-only --no-timing --no-cross-check --workers 1 --steps 3 --warmup 1 > /dev/null 2>&1
  python3 tools/profile_summary.py stats /tmp/pm_$rep gpurun_out/modes_$rep.csv "rep $rep" > /dev/null
  grep -i "latticeTrianglesRow\|latticeVertices" gpurun_out/modes_$rep.csv
done
python bench.py --headline-only --steps 60 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
