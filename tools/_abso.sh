# usage: bash tools/_abso.sh "<bench args>" variantA variantB ... (ab/<name>.so), three interleaved rounds
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
args="$1"; shift
cp mlsgpu_amd/libmlsgpu_hip.so /tmp/orig.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp ab/$v.so mlsgpu_amd/libmlsgpu_hip.so
    python bench.py $args > gpurun_out/abso_$v.json 2> gpurun_out/abso_$v.err
    python - <<P
import json
d=json.loads(open('gpurun_out/abso_$v.json').read().strip().splitlines()[-1])
k=d.get('kernel_ms_per_step',{})
print('$v', d['value'], d['ms_per_step'], {n.split('.')[-2]:round(x,3) for n,x in k.items() if x > 0.4})
P
  done
done
cp /tmp/orig.so mlsgpu_amd/libmlsgpu_hip.so
