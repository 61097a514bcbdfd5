"""The one-process-per-GPU path on real devices: two ranks (sharing GPU 0 on a one-GPU box, gloo for the exchange), each
running ITS share of one cloud's buckets through its own bucket farm into its own welder -- a HostMesher behind the
read-back ring, then a device sink whose meshes never leave HBM -- followed by the one all-gather of
dist_sink.global_prune; and bench.py's N > 1 mode launched the way the driver launches it."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


RANK_CODE = r"""
import os, sys
sys.path[:0] = [%(tests)r, %(root)r, %(oracle)r]
import numpy as np
import torch                      # before the HIP library (tests/conftest.py)
import torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
import mesher_oracle as mo
import oracle_binding as ob
import mlsgpu_amd as m
from mlsgpu_amd import dist_sink, farm, synth
ndev = torch.cuda.device_count()
cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
cloud = np.concatenate([cloud, synth.sphere_cloud(600, (10.0, 10.0, 10.0), 4.0, 1.0, 1.5, seed=3)])
allb, buckets = synth.bucketize(cloud, 96, 32)
mine = farm.rank_share(list(range(len(buckets))), rank, world)
ok = True
results = []
for route in ("host", "device"):
    # host: ship-outs read back through the pinned ring into this rank's HostMesher; device: appended to this rank's
    # device sink, the meshes stay in HBM and only the boundary travels
    ctx = m.Context(rank %% ndev)
    welder = m.HostMesher(0.02) if route == "host" else m.Mesher(ctx, 0.02)
    f = m.BucketFarm([rank %% ndev], max(b.count for b in buckets), workers_per_device=2, max_cells=63,
                     sink=None if route == "host" else welder)
    if route == "host":
        f.set_host_output(8 << 20, welder)
    for i in mine:
        b = buckets[i]
        f.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, rank)
    f.finish()
    f.close()
    n, stats = dist_sink.global_prune(welder, 0.02, dist)
    got = welder.chunk(0) if n else None
    if got is not None and route == "device":
        got = (got["chunk"], got["vertices"], got["triangles"])
    results.append((n, stats, got))
if rank == 0:
    ref = allb.copy()
    everything = []
    for i, b in enumerate(buckets):
        owner = next(r for r in range(world) if i in farm.rank_share(list(range(len(buckets))), r, world))
        batches, _ = ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, max_cells=63, max_swathe=64,
                               mesh_memory=63 * 63 * 2 * 872)
        for g in batches:
            ni = g["num_internal"]
            everything.append(dict(chunk=owner, vertices=g["vertices"], num_internal=ni, keys=g["keys"][ni:], triangles=g["triangles"]))
    exp, exp_stats = mo.mesh_sink(everything, 0.02)
    ev, et = [(v_, t_) for c, v_, t_ in exp if c == 0][0]
    for n, stats, got in results:
        ok = ok and all(stats[k] == exp_stats[k] for k in exp_stats) and stats["kept_components"] < stats["components"]
        ok = ok and n == 1 and mo.isomorphic(got[1], got[2], ev, et)
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 3)
"""


def _launch(n, args, env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=900)


def test_two_ranks_weld_one_cloud(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(RANK_CODE % dict(tests=os.path.join(ROOT, "tests"), root=ROOT, oracle=os.path.join(ROOT, "oracle")))
    out = _launch(2, [str(script)])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]


def test_bench_two_ranks_one_sharded_cloud():
    """bench.py as the driver starts it for N = 2 (here both ranks on GPU 0, gloo): the cfg4 slab family at 2 %% of the
    splat count; the line reports two ranks with 25 buckets each of ONE cloud, and refuses a mismatching --gpus."""
    env = {"MLSGPU_BENCH_BACKEND": "gloo"}
    out = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1",
                      "--no-timing"], env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["per_rank"]["buckets"] == [25, 25]
    assert d["config"]["voxels_per_step"] == 1023 * 1023 * 255
    assert "cfg4" in d["config"]["workload"] and d["value"] > 0
    # round 3: what limits N GPUs is in the line -- the per-GPU reference measured in the same run, the transfer-inclusive
    # region with the cross-rank weld, and the reference's one-process shape (host-fed and device-fed)
    assert d["per_gpu_reference"]["value"] > 0 and 0 < d["scaling_efficiency"] < 2
    ti = d["transfer_inclusive"]["device_sink_global_weld"]
    assert ti["value"] > 0 and ti["h2d_GB_per_step"] > 0 and ti["whole_job"]["kept_components"] >= 1
    sp = d["single_process"]
    assert sp["devices"] == [0, 0] and sp["buckets_per_pass"] == 50
    assert sp["host_fed"]["value"] > 0 and sp["device_fed"]["value"] > 0 and sp["in_flight_max"] >= 2
    assert sum(sp["buckets_per_device_device_fed"]) == 50 * sp["device_fed_passes"] and min(sp["buckets_per_device_device_fed"]) > 0
    # without a launcher bench.py starts the ranks itself; with a launcher of the wrong size it refuses
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.02", "--steps", "1",
                          "--warmup", "0", "--no-timing"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 2
    out = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--scale", "0.02", "--steps", "1", "--warmup", "0"], env)
    assert out.returncode != 0 and "does not match WORLD_SIZE" in (out.stdout + out.stderr)


def test_driver_shaped_call_finishes_in_time():
    """`bench.py --gpus 2 --steps 20 --warmup 5` exactly as the driver launches it, at FULL size (two slabs of cfg4: 200 M-splat
    cloud per rank, 25 buckets each), both ranks on this box's one GPU: the line arrives within 300 s whatever the secondary
    legs do -- they share a wall-clock budget (--leg-budget-s, 150 s by default at N > 1) and a leg that does not fit is
    skipped and named in leg_errors -- and the headline with the in-run per-GPU reference is on stderr before any leg starts."""
    import time
    env = {"MLSGPU_BENCH_BACKEND": "gloo"}
    t0 = time.time()
    out = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"], env)
    elapsed = time.time() - t0
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert elapsed < 300, elapsed
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["per_rank"]["buckets"] == [25, 25] and d["output_digest"]["ok"]
    assert d["per_gpu_reference"]["value"] > 0 and 0 < d["scaling_efficiency"] < 2
    early = [l for l in out.stderr.splitlines() if l.startswith("bench.py headline before the secondary legs: ")]
    assert len(early) == 1
    e = json.loads(early[0].split(": ", 1)[1])
    assert e["value"] == d["value"] and e["scaling_efficiency"] == d["scaling_efficiency"]
    # every secondary leg either reported or was skipped by name
    for leg in ("transfer_inclusive", "single_process"):
        assert leg in d or any(k.startswith(leg) for k in d.get("leg_errors", {})), leg
    # and --legs none is the headline alone
    out = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1",
                      "--legs", "none"], env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "transfer_inclusive" not in d and "single_process" not in d and d["per_gpu_reference"]["value"] > 0


def test_bench_rccl_code_path_with_one_rank():
    """The N > 1 code path of bench.py over RCCL (`nccl` backend: process group with a device id, float64 / int64 all-reduces
    on the GPU, all_gather_object, the gloo side group the waiting ranks park on, every N > 1 leg) with ONE rank on the box's
    GPU -- the shared-GPU runs above use gloo, and a one-GPU box cannot hold two RCCL ranks."""
    env = {"MLSGPU_BENCH_FORCE_DIST": "1", "MLSGPU_BENCH_BACKEND": "nccl"}
    out = _launch(1, [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "cfg4slab", "--scale", "0.02", "--steps", "2",
                      "--warmup", "1", "--no-timing"], env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and "debug_forced_dist" in d and d["per_rank"]["buckets"] == [25]
    assert d["per_gpu_reference"]["value"] > 0 and "leg_errors" not in d
    assert d["transfer_inclusive"]["device_sink_global_weld"]["value"] > 0
    assert d["single_process"]["host_fed"]["value"] > 0 and d["single_process"]["device_fed"]["value"] > 0


def test_bench_greedy_dispatch_two_groups_on_one_gpu():
    """`bench.py --gpus 2 --dispatch greedy`: ONE process, two device groups (both on GPU 0 here: MLSGPU_TEST_DEVICES=0,0), the
    whole cfg4 cloud's 125 buckets handed out by the reference's rule, every bucket against its pin, host-fed and
    device-fed; both groups get work."""
    env = dict(os.environ, MLSGPU_TEST_DEVICES="0,0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dispatch", "greedy", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["output_digest"]["pinned"] and line["output_digest"]["ok"] and line["output_digest"]["totals_ok"]
    assert [d["device"] for d in line["per_device"]] == [0, 0]
    assert sum(d["buckets"] for d in line["per_device"]) == 2 * 125 and min(d["buckets"] for d in line["per_device"]) > 0
    assert "debug_shared_gpu" in line and "error" not in (line["device_fed"] or {})
