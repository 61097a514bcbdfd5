"""Committed fixtures (tests/golden/oracle_golden.json): the oracle must keep reproducing them, and the HIP path
must produce the same digests."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

GOLDEN = json.load(open(os.path.join(HERE, "golden", "oracle_golden.json")))


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_oracle_reproduces_golden(name):
    assert make_golden.run_case(make_golden.CASES[name]) == GOLDEN[name]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_hip_matches_golden(name):
    import mlsgpu_amd as m
    spec = make_golden.CASES[name]
    gold = GOLDEN[name]
    cloud = make_golden.make_cloud(spec["cloud"])
    ctx = m.Context(0)
    w = m.Worker(ctx, len(cloud), max_cells=spec["max_cells"], shape=spec.get("shape", 0),
                 boundary_limit=spec.get("boundary_limit", 1.0), max_swathe=spec.get("max_swathe", 0),
                 mesh_memory=spec.get("mesh_memory", 0))
    batches = w.process(m.DeviceBuffer(ctx, array=cloud), 0, len(cloud), spec["low"], spec["nv"])
    assert [len(b["vertices"]) for b in batches] == gold["vertices"]
    assert [len(b["triangles"]) for b in batches] == gold["triangles"]
    assert [b["num_internal"] for b in batches] == gold["internal"]
    assert [int(x) for x in batches[0]["vertices"][0].view(np.uint32)] == gold["first_vertex_bits"]
    assert make_golden.digest_batches(batches) == gold["digest"]
    del w
    ctx.close()


def test_partition_and_sink_oracles_reproduce_golden():
    assert make_golden.partition_case() == GOLDEN["_partition_random4"]
    assert make_golden.sink_case() == GOLDEN["_sink_prune"]
