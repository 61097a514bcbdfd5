mkdir -p gpurun_out/r03
(time timeout 900 python -m pytest tests/test_gpu_mls.py tests/test_gpu_bucket.py tests/test_dist_gpu.py -x -q -m gpu) > gpurun_out/r03/t4.log 2>&1
tail -4 gpurun_out/r03/t4.log
for v in 2 3 2 3; do
  python bench.py --headline-only --steps 40 --variant $v > gpurun_out/r03/ab_v$v.json 2> gpurun_out/r03/ab_v$v.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r03/ab_v$v.json').read().strip().splitlines()[-1])
print('variant $v', d['value'], d['ms_per_step'], d['kernel_ms_per_step']['kernel.mls.processCorners.time'], d['roofline']['avg_launch_ms'])
P
done
python bench.py --headline-only --steps 40 --variant 3 --dist shells > gpurun_out/r03/ab_v3_shells.json 2>/dev/null
python bench.py --headline-only --steps 40 --variant 2 --dist shells > gpurun_out/r03/ab_v2_shells.json 2>/dev/null
python - <<P
import json
for v in (2,3):
    d=json.loads(open('gpurun_out/r03/ab_v%d_shells.json'%v).read().strip().splitlines()[-1])
    print('shells variant', v, d['value'], d['ms_per_step'], d['kernel_ms_per_step']['kernel.mls.processCorners.time'])
P
rm -f gpurun_out/r03/sq.csv
bash tools/sq_counters.sh 2 "hit lists (variant 2)" gpurun_out/r03/sq.csv
bash tools/sq_counters.sh 3 "hit masks (variant 3)" gpurun_out/r03/sq.csv
cat gpurun_out/r03/sq.csv
export MLSGPU_WRITE_GOLDEN=1
(time timeout 1500 python -m pytest tests/test_gpu_configs.py::test_cfg5_full_shape_from_files -x -q -m gpu) > gpurun_out/r03/t1.log 2>&1
tail -5 gpurun_out/r03/t1.log
(time MLSGPU_CFG5_SPLATS=125000000 timeout 900 python -m pytest tests/test_gpu_configs.py::test_cfg5_full_shape_from_files -x -q -m gpu) > gpurun_out/r03/t2.log 2>&1
tail -5 gpurun_out/r03/t2.log
unset MLSGPU_WRITE_GOLDEN
(time timeout 1200 python bench.py --workload cfg5 --steps 3 --warmup 1) > gpurun_out/r03/bench_cfg5.json 2> gpurun_out/r03/bench_cfg5.err
tail -c 1000 gpurun_out/r03/bench_cfg5.err; head -c 400 gpurun_out/r03/bench_cfg5.json
rm -f /dev/shm/mlsgpu_cfg5_*
