"""The synthetic inputs of the BASELINE configurations: the device-side generators used by bench.py and the full-size
tests produce the same bits as the numpy generators the parity tests use (torch on CPU here; the same code runs on
the GPU)."""
import numpy as np
import pytest


def test_uniform_cloud_device_is_bit_identical():
    import torch
    from mlsgpu_amd import synth
    for cfg, scale in (("cfg3", 0.0004), ("cfg4", 0.0001)):
        ref, g = synth.make_cloud(cfg, scale=scale)
        got, g2 = synth.make_cloud_device(cfg, torch.device("cpu"), scale=scale)
        assert g == g2 and got.shape == (len(ref), 8)
        np.testing.assert_array_equal(got.numpy().view(np.uint32).ravel(), ref.view(np.uint32).ravel())
    # chunked generation and a non-zero first id address the same stream
    a = synth.uniform_cloud_device(5000, 511.0, 2.0, 3.0, synth.cloud_seed("cfg3"), torch.device("cpu"), chunk=1024)
    b = synth.uniform_cloud(5000, 511.0, 2.0, 3.0, synth.cloud_seed("cfg3"))
    np.testing.assert_array_equal(a.numpy().view(np.uint32).ravel(), b.view(np.uint32).ravel())
    c = synth.uniform_cloud_device(100, 511.0, 2.0, 3.0, synth.cloud_seed("cfg3"), torch.device("cpu"), first=4900)
    np.testing.assert_array_equal(c.numpy().view(np.uint32).ravel(), b[4900:].view(np.uint32).ravel())


def test_bucketize_device_matches_host_bucketize():
    import torch
    from mlsgpu_amd import synth
    cloud, g = synth.make_cloud("cfg2", scale=0.004)
    allb, buckets = synth.bucketize(cloud, g, 100)
    boxes = synth.grid_buckets((g, g, g), 100)
    assert [(b.low, b.num_vertices) for b in buckets] == boxes
    t = torch.from_numpy(cloud.view(np.float32).reshape(-1, 8).copy())
    got, gb = synth.bucketize_device(t, boxes)
    np.testing.assert_array_equal(got.numpy().view(np.uint32).ravel(), allb.view(np.uint32).ravel())
    assert [(b.low, b.num_vertices, b.first, b.count) for b in gb] == [(b.low, b.num_vertices, b.first, b.count) for b in buckets]
    # a rank's share is the same buckets, re-based
    share, sb = synth.bucketize_device(t, boxes[5:9])
    first = buckets[5].first
    n = sum(b.count for b in buckets[5:9])
    np.testing.assert_array_equal(share.numpy().view(np.uint32).ravel(), allb[first:first + n].view(np.uint32).ravel())
    assert [b.first for b in sb] == [b.first - first for b in buckets[5:9]]


def test_slab_split_is_balanced():
    from mlsgpu_amd import synth
    for n in (1, 2, 4, 8):
        boxes = synth.grid_buckets((1024, 1024, 128 * n), 255, runs=(0, 0, n))
        assert len(boxes) == 25 * n
        per_rank = [boxes[25 * r:25 * (r + 1)] for r in range(n)]
        for r, share in enumerate(per_rank):
            assert len({b[0][2] for b in share}) == 1                 # one z-run per rank
        cells = sum((nv[0] - 1) * (nv[1] - 1) * (nv[2] - 1) for _, nv in boxes)
        assert cells == 1023 * 1023 * (128 * n - 1)
