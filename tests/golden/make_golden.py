#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_golden.json from the CPU oracle.

The reference holds no golden meshes for this path (all its expectations are analytic or
structural, and those are restated in tests/test_oracle_*.py).  These fixtures are therefore the ORACLE's own
outputs on seeded inputs -- counts, SHA-256 digests and a few sampled values -- committed so that a change in
the oracle, the compiler flags or the floating-point contract shows up as a diff, and so that the HIP path can
be checked against fixed numbers as well as against a live oracle run.

    python tests/golden/make_golden.py        # rewrites oracle_golden.json
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

CASES = {
    # name: (cloud generator args, bucket low extent, number of vertices, oracle parameters)
    "cfg1_sphere_64": dict(cloud=("cfg1", 1.0), low=(0, 0, 0), nv=(64, 64, 64), max_cells=63),
    "offset_ragged": dict(cloud=("sphere", 30000, (70.0, 61.0, 52.0), 17.0, 1.0, 2.5, 99), low=(45, 37, 30),
                          nv=(51, 46, 44), max_cells=50, max_swathe=8, mesh_memory=50 * 50 * 872),
    "plane_cap": dict(cloud=("sphere", 20000, (30.0, 30.0, 30.0), 40.0, 1.5, 2.5, 5), low=(0, 0, 0), nv=(64, 64, 64),
                      max_cells=63, shape=1, boundary_limit=1.5),
}


def make_cloud(spec):
    from mlsgpu_amd import synth
    if spec[0] == "cfg1":
        return synth.make_cloud("cfg1", scale=spec[1])[0]
    _, n, c, big_r, r_lo, r_hi, seed = spec
    return synth.sphere_cloud(n, c, big_r, r_lo, r_hi, seed=seed)


def digest_batches(batches):
    h = hashlib.sha256()
    for b in batches:
        h.update(np.uint64(len(b["vertices"])).tobytes())
        h.update(np.uint64(b["num_internal"]).tobytes())
        h.update(np.ascontiguousarray(b["vertices"]).tobytes())
        h.update(np.ascontiguousarray(b["triangles"]).tobytes())
        h.update(np.ascontiguousarray(b["keys"][b["num_internal"]:]).tobytes())
    return h.hexdigest()


def run_case(spec):
    import oracle_binding as ob
    cloud = make_cloud(spec["cloud"])
    kw = dict(max_cells=spec["max_cells"], shape=spec.get("shape", 0), boundary_limit=spec.get("boundary_limit", 1.0))
    kw["max_swathe"] = spec.get("max_swathe", (spec["max_cells"] + 8) // 8 * 8)
    kw["mesh_memory"] = spec.get("mesh_memory", spec["max_cells"] ** 2 * 2 * 872)
    batches, st = ob.bucket(cloud.copy(), 0, len(cloud), spec["nv"], spec["low"], **kw)
    first = batches[0]
    return dict(
        splats=len(cloud), batches=len(batches),
        vertices=[len(b["vertices"]) for b in batches], triangles=[len(b["triangles"]) for b in batches],
        internal=[b["num_internal"] for b in batches],
        listed=st["listed"], hits=st["hits"], occupied=st["occupied"], commands=st["commands"],
        digest=digest_batches(batches),
        first_vertex_bits=[int(x) for x in first["vertices"][0].view(np.uint32)],
        first_triangle=[int(x) for x in first["triangles"][0]],
    )


def partition_case():
    """Bucket::bucket oracle on a seeded cloud: leaves as extents + SHA-256 of the member ids."""
    from bucket_checks import random_case
    splats, grid, p = random_case(4)
    leaves = ob_module().bucket_partition(splats, grid["reference"], grid["spacing"], grid["extents"], p["max_splats"],
                                          p["max_cells"], p["chunk_cells"], p["micro_cells"], p["max_split"])
    h = hashlib.sha256()
    for leaf in leaves:
        h.update(np.asarray(leaf["extents"], np.int64).tobytes())
        h.update(np.asarray(leaf["chunk"], np.uint64).tobytes())
        h.update(np.ascontiguousarray(leaf["ids"], np.uint64).tobytes())
    return dict(leaves=len(leaves), members=int(sum(len(l["ids"]) for l in leaves)), digest=h.hexdigest(),
                first_extents=[int(v) for v in leaves[0]["extents"]])


def sink_case():
    """Mesh-sink oracle on the reference's prune fixture: statistics and canonical mesh digest."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import mesher_oracle as mo
    from mesher_cases import CASES as MC
    out, stats = mo.mesh_sink(MC["prune"]["meshes"], MC["prune"]["prune"])
    v, t = mo.canonical(out[0][1], out[0][2])
    h = hashlib.sha256(np.ascontiguousarray(v).tobytes() + np.ascontiguousarray(t, np.int64).tobytes())
    return dict(stats=stats, vertices=len(v), triangles=len(t), digest=h.hexdigest())


def ob_module():
    import oracle_binding as ob
    return ob


def main():
    out = {name: run_case(spec) for name, spec in sorted(CASES.items())}
    out["_partition_random4"] = partition_case()
    out["_sink_prune"] = sink_case()
    with open(os.path.join(HERE, "oracle_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps({k: (v["digest"][:16], v.get("vertices"), v.get("triangles")) for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
