"""Shared helpers of the -m gpu tests."""
import numpy as np
import pytest


def get_ctx():
    import mlsgpu_amd
    return mlsgpu_amd.Context(0)


@pytest.fixture(scope="module")
def ctx():
    c = get_ctx()
    yield c
    c.close()


def assert_batches_equal(got, exp):
    """Bit-exact comparison of ship-out batches (HIP path vs oracle)."""
    assert len(got) == len(exp), (len(got), len(exp))
    for g, e in zip(got, exp):
        assert g["num_internal"] == e["num_internal"]
        assert g["vertices"].shape == e["vertices"].shape
        assert g["triangles"].shape == e["triangles"].shape
        np.testing.assert_array_equal(g["vertices"].view(np.uint32), e["vertices"].view(np.uint32))
        np.testing.assert_array_equal(g["triangles"], e["triangles"])
        ni = e["num_internal"]
        np.testing.assert_array_equal(g["keys"][ni:], e["keys"][ni:])
