#!/bin/bash
# Runs ON THE GPU BOX (gpurun): bench.py under several builds of the library (ab/<name>.so, see tools/ab_kernel.sh), three
# interleaved rounds on the same box:  gpurun -- 'bash tools/ab_bench.sh "--headline-only --steps 60 --warmup 5" base variant'
# AB_TESTS="tests/test_gpu_marching.py ..." runs those GPU tests first (with the library that is in place).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
args="$1"; shift
cp mlsgpu_amd/libmlsgpu_hip.so /tmp/orig.so
if [ -n "$AB_TESTS" ]; then python -m pytest $AB_TESTS -m gpu -x -q 2>&1 | tail -2; fi
for rep in 1 2 3; do
  for v in "$@"; do
    cp ab/$v.so mlsgpu_amd/libmlsgpu_hip.so
    python bench.py $args > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
    python - <<P
import json
d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1])
k=d.get('kernel_ms_per_step',{})
print('$v', d['value'], d['ms_per_step'], {n.split('.')[-2]:round(x,3) for n,x in k.items() if x > 0.4})
P
  done
done
cp /tmp/orig.so mlsgpu_amd/libmlsgpu_hip.so
