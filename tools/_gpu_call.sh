mkdir -p gpurun_out/r03
(timeout 600 python -m pytest tests/test_gpu_tree.py tests/test_gpu_bucket.py tests/test_gpu_marching.py -x -q -m gpu) 2>&1 | tail -2
for w in 1 4; do for sp in 0 1 0 1; do
MLSGPU_HIP_MARCHING_SPECULATE=$sp python bench.py --headline-only --no-timing --workers $w --steps 40 > gpurun_out/r03/h.json 2> gpurun_out/r03/h.err
python - <<P
import json
d=json.loads(open('gpurun_out/r03/h.json').read().strip().splitlines()[-1])
print('workers $w speculate $sp', d['value'], d['ms_per_step'])
P
done; done
for sp in 0 1; do
MLSGPU_HIP_MARCHING_SPECULATE=$sp python bench.py --steps 20 --no-timing --no-cpu-baseline --no-transfer --no-shells --no-sink > gpurun_out/r03/h.json 2> gpurun_out/r03/h.err
python - <<P
import json
d=json.loads(open('gpurun_out/r03/h.json').read().strip().splitlines()[-1])
print('partition leg speculate $sp', d['device_partition']['pipeline_ms_per_step'])
P
done
