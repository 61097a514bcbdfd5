"""Bucket farm plumbing shared by bench.py and the tests: how buckets / clouds are split over ranks and
device workers, and how per-rank timings and counts are combined.  No data-path collective exists on this
path (buckets are independent); torch.distributed only carries a barrier and two reductions.
"""


def worker_share(items, k, nworkers):
    """Buckets of device worker k of `nworkers` on one GPU: k, k + nworkers, ... (greedy in arrival order,
    like DeviceWorkerGroup's queue, src/worker_group.h)."""
    return items[k::nworkers]


def rank_share(items, rank, world):
    """Strong-scaling split of ONE bucket stream over ranks: contiguous runs, sizes differing by at most one."""
    n = len(items)
    base, extra = divmod(n, world)
    first = rank * base + min(rank, extra)
    return items[first:first + base + (1 if rank < extra else 0)]


def combine(elapsed_s, units, dist=None, device=None):
    """Whole-job figures from per-rank ones: the MAX of the elapsed time and the SUM of the units processed.
    `dist` is torch.distributed (already initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), int(units), 1
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([int(units)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item()), dist.get_world_size()


def throughput(units, elapsed_s):
    """Mega-units per second."""
    return units / elapsed_s / 1e6


def leaf_geometry(leaf, ext):
    """(low extent relative to the grid, corners per axis) of a Bucket::bucket leaf."""
    low = [leaf["extents"][2 * a] - ext[2 * a] for a in range(3)]
    nv = [leaf["extents"][2 * a + 1] - leaf["extents"][2 * a] + 1 for a in range(3)]
    return low, nv


def partition_to_farm(ctx, bucket_farm, device, raw, num_splats, reference, spacing, ext, params, chunk_of=None,
                      keep=None):
    """A cloud resident on `device` -> Bucket::bucket on the device (mlsgpu_hip_bucket) -> every leaf handed to the
    farm's device path while its id list is valid (gather + transform kernel into a device item, peer copy for another
    GPU's group), as CopyGroup + BucketLoader do with host buckets (src/workers.cpp:377-418).  chunk_of(leaf number) ->
    chunk id (default: the leaf number, one chunk per bucket); keep(leaf number) -> False skips a leaf (another rank's).
    Returns the list of leaves (all of them, skipped ones included).  The caller calls bucket_farm.finish()."""
    from . import binding as mb
    count = [0]
    def leaf_work(leaf, d_ids):
        i = count[0]
        count[0] += 1
        if keep is not None and not keep(i):
            return None
        low, nv = leaf_geometry(leaf, ext)
        # only enqueued: the bucketer orders the reuse of the id list behind the returned event, on the GPU
        return bucket_farm.submit_device(device, raw, d_ids, leaf["num_splats"], reference, spacing, ext, low, nv,
                                         i if chunk_of is None else chunk_of(i), wait=False)
    return mb.bucket_cloud(ctx, raw, num_splats, reference, spacing, ext, on_bucket=leaf_work, **params)


def leaf_cells(leaf):
    e = leaf["extents"]
    return (e[1] - e[0]) * (e[3] - e[2]) * (e[5] - e[4])


def node_cpus(node, root=None):
    """CPUs of a NUMA node from sysfs ([] when the node is not listed)."""
    import os
    root = root or os.environ.get("MLSGPU_HIP_SYSFS_ROOT") or "/sys"
    try:
        text = open("%s/devices/system/node/node%d/cpulist" % (root, node)).read().strip()
    except OSError:
        return []
    out = []
    for part in text.split(","):
        if part:
            lo, _, hi = part.partition("-")
            out += range(int(lo), int(hi or lo) + 1)
    return out


def device_node_out_of_process(device):
    """The NUMA node of HIP device `device`, asked of a CHILD process (same environment, same device enumeration): this
    process's HIP runtime stays untouched, so that it can be bound before the runtime starts its own threads."""
    import subprocess
    import sys
    code = ("import ctypes,sys\n"
            "hip=ctypes.CDLL('libamdhip64.so')\n"
            "b=ctypes.create_string_buffer(64)\n"
            "sys.exit(9) if hip.hipDeviceGetPCIBusId(b,64,%d)!=0 else print(b.value.decode().lower())" % int(device))
    try:
        bdf = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1]
        import os
        root = os.environ.get("MLSGPU_HIP_SYSFS_ROOT") or "/sys"
        return int(open("%s/bus/pci/devices/%s/numa_node" % (root, bdf)).read())
    except Exception:   # noqa: BLE001 - no GPU, no sysfs entry: unknown
        return -1


def bind_process_to_device_node(device, before_hip=False):
    """One process per GPU: the process -- every thread it starts from now on, every first touch of host memory -- on the
    CPUs of the NUMA node its GPU hangs off (what `numactl --cpunodebind` does for a launcher that knows the topology;
    torch.distributed.run does not).  Call it before anything is allocated; with before_hip=True even before the HIP
    runtime is initialised (the node is asked of a child process), so that the runtime's own threads -- the ones that
    wake a host thread waiting for an event -- start on the GPU's socket too.  Returns what was done, for the bench line:
    {"numa_nodes", "gpu_node", "bound", "cpus"}.  A one-node machine, an unknown node or an affinity mask that excludes the
    node's CPUs leaves the process alone.  The library's own threads (the farm's copy sides, device workers, the host
    welder) place themselves per GPU either way (csrc/placement.hpp)."""
    import os
    root = os.environ.get("MLSGPU_HIP_SYSFS_ROOT") or "/sys"
    nodes = 0
    while os.path.exists("%s/devices/system/node/node%d/cpulist" % (root, nodes)):
        nodes += 1
    forced = os.environ.get("MLSGPU_HIP_DEVICE_NODES")
    if forced:
        listed = [int(x) for x in forced.split(",") if x.strip()]
        node = listed[int(device)] if int(device) < len(listed) else -1
    elif before_hip:
        node = device_node_out_of_process(device) if nodes >= 2 else -1
    else:
        from . import binding as mb
        node = int(mb.lib().mlsgpu_hip_device_node(int(device)))
    out = {"numa_nodes": nodes, "gpu_node": node, "bound": False, "cpus": len(os.sched_getaffinity(0)),
           "before_hip_runtime": bool(before_hip)}
    if nodes < 2 or node < 0:
        return out
    want = set(node_cpus(node)) & os.sched_getaffinity(0)
    if not want:
        return out
    os.sched_setaffinity(0, want)
    out.update(bound=True, cpus=len(want))
    return out


def slab_verdicts(pins, world, rank, digest, vertices, triangles, dist=None, device=None):
    """Every rank of the sharded cfg4 workload holds the digest of its slab's meshes against the pin of THAT slab (slab
    `rank` as an inner slab, or as the job's last one: synth.slab_variant); the verdicts and digests are gathered, so that
    rank 0 can report all of them and every rank knows whether ANY slab differed.  `pins`: the "slabs" table of
    tests/golden/cfg4slab_uniform.json.  Returns dict(ok=[...], digests=[...], expected=[...], all_ok=bool)."""
    from . import synth
    pin = (pins or {}).get(str(rank), {}).get(synth.slab_variant(world, rank))
    mine_ok = pin is not None and digest == pin["digest"] and vertices == pin["vertices"] and triangles == pin["triangles"]
    ok, digests = [bool(mine_ok)], [digest]
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch
        d = torch.zeros((world, 3), dtype=torch.int64, device=device)
        d[rank, 0], d[rank, 1], d[rank, 2] = int(mine_ok), int(digest[:8], 16), int(digest[8:], 16)
        dist.all_reduce(d)
        d = d.cpu().numpy()
        ok = [bool(x) for x in d[:, 0]]
        digests = ["%08x%08x" % (int(a), int(b)) for a, b in d[:, 1:3]]
    expected = [(pins or {}).get(str(r), {}).get(synth.slab_variant(world, r), {}).get("digest") for r in range(world)]
    return dict(ok=ok, digests=digests, expected=expected, all_ok=all(ok))
