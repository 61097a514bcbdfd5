cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in pipe new old pipe new old pipe new old; do
  unset MLSGPU_HIP_TRI_PIPE MLSGPU_HIP_TRI_OLD
  if [ $v = pipe ]; then export MLSGPU_HIP_TRI_PIPE=1; fi
  if [ $v = old ]; then export MLSGPU_HIP_TRI_OLD=1; fi
  python bench.py --headline-only --steps 60 --warmup 5 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - <<P
import json
d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1])
k=d.get('kernel_ms_per_step',{})
print('$v', d['value'], d['ms_per_step'], {n:round(x,3) for n,x in k.items() if 'generateElements' in n})
P
done
