"""Placement (mlsgpu_amd/csrc/placement.hpp): which copy side serves which GPU, where threads are bound.  The machine is a
fake one -- a sysfs tree in a temporary directory with two NUMA nodes and eight GPUs -- so the assignment an 8-GPU,
two-socket node would get is checked here, on a CPU-only box.  The reference places nothing (src/workers.cpp:320-351)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def plan(device_nodes, num_nodes):
    import mlsgpu_amd as m
    dn = np.array(device_nodes, np.int32)
    side = np.zeros(len(dn), np.int32)
    node_of_side = np.full(16, -9, np.int32)
    n = np.zeros(1, np.uint32)
    m.binding.check(m.lib().mlsgpu_hip_plan_copy_sides(dn.ctypes.data, len(dn), num_nodes, side.ctypes.data,
                                                       node_of_side.ctypes.data, n.ctypes.data))
    return side.tolist(), node_of_side[:int(n[0])].tolist()


def test_plan_one_side_per_socket():
    # an 8 x MI355X node: GPUs 0-3 behind socket 0, 4-7 behind socket 1
    assert plan([0, 0, 0, 0, 1, 1, 1, 1], 2) == ([0, 0, 0, 0, 1, 1, 1, 1], [0, 1])
    # sides are numbered by first appearance: device 0's side is side 0 whatever its node
    assert plan([1, 0, 1, 0], 2) == ([0, 1, 0, 1], [1, 0])
    # a GPU whose node sysfs does not give goes with the first side
    assert plan([1, -1, 0], 2) == ([0, 0, 1], [1, 0])
    # one NUMA node (or none reported): one unbound side
    assert plan([0, 0], 1) == ([0, 0], [-1])
    assert plan([-1, -1, -1], 2) == ([0, 0, 0], [-1])
    # the one-GPU box of this pool: GPU on node 1 of 2
    assert plan([1], 2) == ([0], [1])


FAKE = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import mlsgpu_amd as m
L = m.lib()
n = np.zeros(1, np.uint32); per = np.zeros(16, np.uint32)
m.binding.check(L.mlsgpu_hip_topology(n.ctypes.data, per.ctypes.data))
out = {"nodes": int(n[0]), "cpus": per[:int(n[0])].tolist(), "device_nodes": [L.mlsgpu_hip_device_node(d) for d in range(9)]}
before = sorted(os.sched_getaffinity(0))
out["bound1"] = L.mlsgpu_hip_bind_thread_to_node(1)
out["affinity1"] = sorted(os.sched_getaffinity(0))
out["bound0"] = L.mlsgpu_hip_bind_thread_to_node(0)
out["affinity0"] = sorted(os.sched_getaffinity(0))
out["bound_unknown"] = L.mlsgpu_hip_bind_thread_to_node(-1)
out["affinity_unknown"] = sorted(os.sched_getaffinity(0))
out["before"] = before
out["pool_bad"] = L.mlsgpu_hip_test_copy_pool(4, 200, 9 << 20, 1)
# a welder bound to node 1 welds what an unbound one does
rng = np.random.default_rng(5)
v = rng.random((2000, 3)).astype(np.float32); t = rng.integers(0, 2000, (4000, 3)).astype(np.uint32); k = np.arange(500, dtype=np.uint64)
res = []
for node in (-1, 1):
    w = m.HostMesher(0.0, threads=3)
    w.set_node(node)
    w.add(0, v, 1500, k, t); w.finalize()
    res.append(w.chunk(0)); w.close()
out["welder_same"] = bool(all(np.array_equal(a, b) for a, b in zip(res[0][1:], res[1][1:])))
print(json.dumps(out))
'''


def test_fake_two_socket_machine(tmp_path):
    """sysfs of a machine with two nodes (CPUs 0-1 and 2-3) and eight GPUs, four per node: the topology read, every GPU's
    node, and a thread bound to a node really runs on that node's CPUs only."""
    avail = sorted(os.sched_getaffinity(0))
    if len(avail) < 4:
        import pytest
        pytest.skip("needs four CPUs")
    a, b = avail[:2], avail[2:4]
    for node, cpus in ((0, a), (1, b)):
        d = tmp_path / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    env = dict(os.environ, MLSGPU_HIP_SYSFS_ROOT=str(tmp_path), MLSGPU_HIP_DEVICE_NODES="0,0,0,0,1,1,1,1")
    out = json.loads(subprocess.check_output([sys.executable, "-c", FAKE % dict(root=ROOT)], env=env).decode().strip().splitlines()[-1])
    assert out["nodes"] == 2 and out["cpus"] == [2, 2]
    assert out["device_nodes"] == [0, 0, 0, 0, 1, 1, 1, 1, -1]
    assert out["bound1"] == 1 and out["affinity1"] == b
    assert out["bound0"] == 1 and out["affinity0"] == a
    assert out["bound_unknown"] == 0 and out["affinity_unknown"] == a          # unknown node: the thread is left alone
    assert out["pool_bad"] == 0 and out["welder_same"]


def test_pci_node_lookup_from_sysfs(tmp_path):
    """Without the override a GPU's node comes from /sys/bus/pci/devices/<bdf>/numa_node; without a GPU the lookup says -1
    and nothing is bound."""
    code = ("import sys; sys.path.insert(0, %r); import mlsgpu_amd as m; L = m.lib(); "
            "print(L.mlsgpu_hip_device_node(0), L.mlsgpu_hip_bind_thread_to_node(L.mlsgpu_hip_device_node(0)))" % ROOT)
    env = dict(os.environ, MLSGPU_HIP_SYSFS_ROOT=str(tmp_path))
    env.pop("MLSGPU_HIP_DEVICE_NODES", None)
    out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:   # noqa: BLE001
        has_gpu = False
    if not has_gpu:
        assert out == ["-1", "0"]


def test_copy_pool_is_exact_at_every_thread_count():
    import mlsgpu_amd as m
    for threads in (1, 2, 3, 8):
        assert m.lib().mlsgpu_hip_test_copy_pool(threads, 60, 11 << 20, -1) == 0
    assert m.lib().mlsgpu_hip_test_copy_pool(4, 5, 0, -1) == 0


def test_copy_pool_copies_the_tail():
    """Sizes whose floor(bytes / parts) is a multiple of the 4 KB chunk granule with a remainder on top (the
    chunk was the rounded FLOOR, so parts x chunk < bytes and the last bytes % parts bytes were never copied), at thread
    counts on both sides of 32 (the farm copies 32-byte splats)."""
    import mlsgpu_amd as m
    for threads, k in ((8, 512), (3, 700), (33, 520), (40, 600)):
        for r in sorted({1, 4, 31, 32, threads - 1}):
            if 0 < r < threads:
                bytes_ = threads * 4096 * k + r
                assert m.lib().mlsgpu_hip_test_copy_pool(threads, 0, bytes_, -1) == 0, (threads, k, r)


BIND = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
from mlsgpu_amd import farm
out = {"before": sorted(os.sched_getaffinity(0))}
out["r0"] = farm.bind_process_to_device_node(0, before_hip=True)
out["after0"] = sorted(os.sched_getaffinity(0))
os.sched_setaffinity(0, out["before"])
out["r5"] = farm.bind_process_to_device_node(5, before_hip=True)
out["after5"] = sorted(os.sched_getaffinity(0))
os.sched_setaffinity(0, out["before"])
out["r8"] = farm.bind_process_to_device_node(8, before_hip=True)      # no such GPU in the table: left alone
out["after8"] = sorted(os.sched_getaffinity(0))
print(json.dumps(out))
'''


def test_rank_binds_itself_to_its_gpus_socket(tmp_path):
    """bench.py's ranks: rank r (GPU r) ends up on the CPUs of the node GPU r hangs off -- GPUs 0-3 on node 0, 4-7 on node 1
    of the fake machine -- before HIP is touched; an unknown GPU leaves the process where it was."""
    avail = sorted(os.sched_getaffinity(0))
    if len(avail) < 4:
        import pytest
        pytest.skip("needs four CPUs")
    a, b = avail[:2], avail[2:4]
    for node, cpus in ((0, a), (1, b)):
        d = tmp_path / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    env = dict(os.environ, MLSGPU_HIP_SYSFS_ROOT=str(tmp_path), MLSGPU_HIP_DEVICE_NODES="0,0,0,0,1,1,1,1")
    out = json.loads(subprocess.check_output([sys.executable, "-c", BIND % dict(root=ROOT)], env=env).decode().strip().splitlines()[-1])
    assert out["r0"]["bound"] and out["r0"]["gpu_node"] == 0 and out["r0"]["numa_nodes"] == 2 and out["after0"] == a
    assert out["r5"]["bound"] and out["r5"]["gpu_node"] == 1 and out["after5"] == b and out["r5"]["cpus"] == 2
    assert not out["r8"]["bound"] and out["after8"] == out["before"]
