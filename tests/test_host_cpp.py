"""The C++ mirror of the reference's class surface (mlsgpu_amd/host/mlsgpu_hip.hpp) and the example host."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(tmp_path, name="host_bucket"):
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "mlsgpu_amd")
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-O1", "-I", ROOT,
           os.path.join(ROOT, "examples", name + ".cpp"), "-o", exe,
           "-L" + libdir, "-lmlsgpu_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
           "-pthread"]
    subprocess.check_call(cmd)
    return exe


def test_host_header_compiles_and_links(tmp_path):
    """SplatTreeCL / MlsFunctor / Marching / DeviceWorkerGroup over the C-ABI: compiles warning-free and links."""
    exe = build_example(tmp_path)
    assert os.path.exists(exe)
    assert os.path.exists(build_example(tmp_path, "host_partition"))
    assert os.path.exists(build_example(tmp_path, "reconstruct"))


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [1, 4])
def test_device_worker_group_matches_oracle(tmp_path, lanes):
    """27 buckets through a 2-thread DeviceWorkerGroup from C++; every ship-out equals the oracle's -- one bucket per work
    item, or four per item taken through the device path in lock-step (DeviceWorkerGroup::setBatch)."""
    import oracle_binding as ob
    from mlsgpu_amd import synth
    exe = build_example(tmp_path)
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    allb.tofile(str(tmp_path / "splats.bin"))
    with open(str(tmp_path / "buckets.txt"), "w") as f:
        for b in buckets:
            f.write("%d %d %d %d %d %d %d %d\n" % ((b.first, b.count) + tuple(b.low) + tuple(b.num_vertices)))
    out = subprocess.check_output([exe, str(tmp_path / "splats.bin"), str(tmp_path / "buckets.txt"),
                                   str(tmp_path / "out.bin"), "2", str(lanes)], timeout=300).decode()
    assert out.startswith("buckets 27")
    raw = np.fromfile(str(tmp_path / "out.bin"), np.uint8)
    got = {}
    pos = 0
    while pos < len(raw):
        chunk, nv, nt, ni = (int(x) for x in raw[pos:pos + 32].view(np.uint64))
        pos += 32
        ne = nv - ni
        keys = raw[pos:pos + 8 * ne].view(np.uint64)
        verts = raw[pos + 8 * ne:pos + 8 * ne + 12 * nv].view(np.float32).reshape(nv, 3)
        tris = raw[pos + 8 * ne + 12 * nv:pos + 8 * ne + 12 * nv + 12 * nt].view(np.uint32).reshape(nt, 3)
        pos += 8 * ne + 12 * nv + 12 * nt
        got.setdefault(chunk, []).append((keys, verts, tris, ni))
    ref = allb.copy()
    nonempty = 0
    for i, b in enumerate(buckets):
        exp, _ = ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, max_cells=63, max_swathe=64,
                           mesh_memory=63 * 63 * 2 * 872)
        mine = got.get(i, [])
        assert len(mine) == len(exp), i
        for (keys, verts, tris, ni), e in zip(mine, exp):
            nonempty += 1
            assert ni == e["num_internal"]
            np.testing.assert_array_equal(verts.view(np.uint32), e["vertices"].view(np.uint32))
            np.testing.assert_array_equal(tris, e["triangles"])
            np.testing.assert_array_equal(keys, e["keys"][ni:])
    assert nonempty > 0


@pytest.mark.gpu
def test_cpp_bucket_matches_oracle(tmp_path):
    """mlsgpu::hip::Bucket::bucket from C++: the same bins as the oracle, DensityError included."""
    import oracle_binding as ob
    from bucket_checks import random_case
    exe = build_example(tmp_path, "host_partition")
    for seed in (1, 4, 11):
        splats, grid, p = random_case(seed)
        splats.tofile(str(tmp_path / "cloud.bin"))
        args = [exe, str(tmp_path / "cloud.bin")] + [repr(float(v)) for v in grid["reference"]] + [repr(grid["spacing"])] \
            + [str(int(v)) for v in grid["extents"]] \
            + [str(p[k]) for k in ("max_splats", "max_cells", "chunk_cells", "micro_cells", "max_split")]
        out = subprocess.check_output(args, timeout=300).decode().split("\n")
        try:
            exp = ob.bucket_partition(splats, grid["reference"], grid["spacing"], grid["extents"], p["max_splats"],
                                      p["max_cells"], p["chunk_cells"], p["micro_cells"], p["max_split"])
        except ob.DensityError as e:
            assert out[0] == "density %d" % e.cell_splats
            continue
        assert out[len(exp)] == "bins %d" % len(exp)
        for line, e in zip(out, exp):
            want = list(e["extents"]) + list(e["chunk"]) + [e["depth"], len(e["ids"]), int(e["ids"].sum())]
            assert [int(v) for v in line.split()] == want


def parse_ply_mesh(path):
    raw = open(path, "rb").read()
    head_end = raw.index(b"end_header\n") + 11
    head = raw[:head_end].decode("ascii").split("\n")
    nv = int([l for l in head if l.startswith("element vertex")][0].split()[2])
    nt = int([l for l in head if l.startswith("element face")][0].split()[2])
    v = np.frombuffer(raw, "<f4", 3 * nv, head_end).reshape(nv, 3)
    faces = np.frombuffer(raw, np.dtype([("n", np.uint8), ("i", "<u4", 3)]), nt, head_end + 12 * nv)
    assert (faces["n"] == 3).all()
    return v, faces["i"]


@pytest.mark.gpu
def test_reconstruct_ply_to_ply(tmp_path):
    """examples/reconstruct: a PLY of splats in, a welded and pruned PLY mesh out; the result equals the oracle chain
    (bucketing oracle -> per-bucket oracle -> mesh-sink oracle) up to vertex / triangle order."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mesher_oracle as mo
    import oracle_binding as ob
    from mlsgpu_amd import synth
    exe = build_example(tmp_path, "reconstruct")
    cloud = synth.shells_cloud(60_000, 63.0, 16.0, 1.5, 2.5, seed=5)          # grid units
    spacing = np.float32(0.5)
    world = cloud.copy()
    world["position"] = world["position"] * spacing + np.float32(3.0)
    world["radius"] = world["radius"] * spacing
    rows = np.zeros(len(world), np.dtype([("p", "<f4", 3), ("n", "<f4", 3), ("r", "<f4")]))
    rows["p"], rows["n"], rows["r"] = world["position"], world["normal"], world["radius"]
    # three input files of unequal size (SplatSet::FileSet), read through a 64 KiB pinned buffer
    cuts = [0, 25_000, 25_001, len(rows)]
    names = []
    for k in range(3):
        part = rows[cuts[k]:cuts[k + 1]]
        head = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % len(part) \
            + "".join("property float32 %s\n" % n for n in ("x", "y", "z", "nx", "ny", "nz", "radius")) + "end_header\n"
        (tmp_path / ("in%d.ply" % k)).write_bytes(head.encode("ascii") + part.tobytes())
        names.append(str(tmp_path / ("in%d.ply" % k)))
    smooth, levels, subsampling, prune, max_splats = 1.5, 4, 3, 0.02, 30000
    tail = [repr(float(spacing)), str(smooth), str(levels), str(subsampling), str(prune), str(max_splats)]
    meshes_out = {}
    for weld, devices in (("device", "0"), ("device", "0,0,0"), ("host", "0,0"), ("device", "0,0,0,0"), ("host", "0")):
        out_ply = tmp_path / ("out_%s_%d.ply" % (weld, len(devices)))
        env = dict(os.environ)
        if devices == "0,0,0":
            env.update(MLSGPU_HIP_FARM_FORCE_PEER="1", MLSGPU_HIP_MESHER_FORCE_PEER="1")      # the cross-GPU routes
        # the fourth run streams the set: at most 25 000 splats of it resident at a time (mlsgpu_hip_bucket_stream)
        extra = ["--hbm-splats", "25000"] if devices == "0,0,0,0" else []
        if (weld, devices) == ("host", "0"):
            extra = ["--tmp-dir", str(tmp_path)]                  # the host welder's blocks in temporary files (OOCMesher's --tmp-dir)
        out = subprocess.check_output([exe, "--devices", devices, "--weld", weld, "--buffer", "65536"] + extra + names
                                      + [str(out_ply)] + tail, timeout=600, env=env).decode()
        assert "files in 3" in out and "files 1" in out and "weld " + weld in out, out
        meshes_out[(weld, devices)] = parse_ply_mesh(out_ply)
    got_v, got_t = meshes_out[("device", "0")]

    # the same chain with the oracles
    splats = world.copy()
    splats["radius"] = np.minimum(splats["radius"], np.float32(np.inf)) * np.float32(smooth)
    # Reader::decode: 1.0 / (radius * radius) with the product in float and the division in double
    splats["quality"] = (1.0 / (splats["radius"] * splats["radius"]).astype(np.float64)).astype(np.float32)
    max_cells = (1 << (levels + subsampling - 1)) - 1
    micro = min(63, max_cells)
    lo = np.floor((splats["position"] - splats["radius"][:, None]).min(axis=0) / spacing).astype(np.int64) // micro * micro
    hi = np.ceil((splats["position"] + splats["radius"][:, None]).max(axis=0) / spacing).astype(np.int64)
    extents = (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
    leaves = ob.bucket_partition(splats, (0, 0, 0), float(spacing), extents, max_splats, max_cells, 0, micro, 1 << 30)
    meshes = []
    for leaf in leaves:
        low = [leaf["extents"][2 * i] - extents[2 * i] for i in range(3)]
        nv = [leaf["extents"][2 * i + 1] - leaf["extents"][2 * i] + 1 for i in range(3)]
        local = splats[leaf["ids"].astype(np.int64)].copy()
        inv = np.float32(1.0) / spacing
        local["position"] = (local["position"] - np.float32(0.0)) * inv - np.array(extents[0::2], np.float32)
        local["radius"] = local["radius"] * inv
        batches, _ = ob.bucket(local, 0, len(local), nv, low, levels=levels, subsampling=subsampling, max_cells=max_cells)
        for g in batches:
            origin = np.float32(0.0) + spacing * np.array(extents[0::2], np.float32)
            verts = np.ascontiguousarray(g["vertices"], np.float32).copy()
            ob.lib().orc_scale_bias(ob._p(verts), len(verts), float(spacing), float(origin[0]), float(origin[1]), float(origin[2]))
            meshes.append(dict(chunk=0, vertices=verts, num_internal=g["num_internal"],
                               keys=g["keys"][g["num_internal"]:], triangles=g["triangles"]))
    exp, stats = mo.mesh_sink(meshes, prune)
    assert len(exp) == 1 and len(got_t) == stats["kept_triangles"] > 10000
    for key, (v, t) in meshes_out.items():          # one GPU, three device groups over the peer routes, host weld: one mesh
        assert mo.isomorphic(v, t, exp[0][1], exp[0][2]), key
