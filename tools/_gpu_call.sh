mkdir -p gpurun_out/r03
bash tools/profile_run.sh r03 > gpurun_out/r03/profile_run.log 2>&1
tail -3 gpurun_out/r03/profile_run.log
for ct in 4 8 16; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-partition --no-sink --copy-threads $ct > gpurun_out/r03/bench_legs_ct$ct.json 2> gpurun_out/r03/bench_legs.err
python - <<P
import json
d=json.loads(open('gpurun_out/r03/bench_legs_ct$ct.json').read().strip().splitlines()[-1])
t=d['transfer_inclusive']; s=d['shells']['transfer_inclusive']
print('ct $ct uniform ship', t['shipouts']['ms_per_step'], 'sink', t['device_sink']['ms_per_step'], t['device_sink']['one_job_alone_ms'], '| shells ship', s['shipouts']['ms_per_step'], 'sink', s['device_sink']['ms_per_step'], s['device_sink']['one_job_alone_ms'])
P
done
