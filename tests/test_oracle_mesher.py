"""Pins the mesh-sink oracle (oracle/mesher_oracle.py) with the vectors of test/test_mesher.cpp:250-1008."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import mesher_oracle as mo  # noqa: E402
from mesher_cases import CASES, mesh  # noqa: E402


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_case(name):
    case = CASES[name]
    out, stats = mo.mesh_sink(case["meshes"], case.get("prune", 0.0))
    assert [c for c, _, _ in out] == [c for c, _, _ in case["expected"]]
    for (_, v, t), (_, ev, et) in zip(out, case["expected"]):
        assert mo.isomorphic(v, t, ev, et), name
    if "stats" in case:
        for k, val in case["stats"].items():
            assert stats[k] == val


def test_order_of_blocks_does_not_matter():
    # test/test_mesher.cpp:497-525 adds the blocks in reverse on the second pass
    case = CASES["simple"]
    out, _ = mo.mesh_sink(case["meshes"][::-1])
    (_, v, t), (_, ev, et) = out[0], case["expected"][0]
    assert mo.isomorphic(v, t, ev, et)


def test_ply_layout():
    v = np.array([[0, 0, 1], [0, 2, 0], [3, 0, 0]], np.float32)
    t = np.array([[0, 1, 2]], np.uint32)
    raw = mo.ply_bytes(v, t, ["mlsgpu version: test"])
    head, body = raw.split(b"end_header\n")
    assert (len(head) + len(b"end_header\n")) % 4 == 0          # padded so that the vertex data is aligned
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\ncomment mlsgpu version: test\nelement vertex 3\n")
    assert b"element face 1\nproperty list uint8 uint32 vertex_indices\ncomment padding:" in head
    assert len(body) == 3 * 12 + 13 and body[36] == 3


def test_isomorphism_checker_rejects_differences():
    v = np.array([[0, 0, 1], [0, 2, 0], [3, 0, 0]], np.float32)
    assert mo.isomorphic(v, [[0, 1, 2]], v[[2, 0, 1]], [[2, 0, 1]])        # relabelled + rotated
    assert not mo.isomorphic(v, [[0, 1, 2]], v, [[0, 2, 1]])                # orientation flipped
