#!/usr/bin/env python3
"""What-ifs for the drain of processCorners (variant 5) on the candidate masks of real tiles.

A library built with -DMLSGPU_MLS5_DUMP=<words> leaves, for a sample of blocks, every tile's 64 candidate masks behind the
work counters (bench.py: MLSGPU_BENCH_MLS_STATS_WORDS / MLSGPU_BENCH_MLS_STATS_FILE).  This script replays them, wave by
wave and round by round, under several ways of walking the masks and prints the iterations a wave spends (one iteration =
one candidate for every lane that still has one: the ~29 vector instructions of the drain's body).

    python3 tools/drain_sim.py gpurun_out/mls_dump_uniform.npz
"""
import sys
from collections import defaultdict

import numpy as np


def load(path):
    w = np.load(path)["words"]
    n = int(w[0]) // 65
    rec = w[1:1 + 65 * n].reshape(n, 65)
    head = rec[:, 0]
    masks = rec[:, 1:].astype(np.uint32)
    key = (head >> 16)          # lane of the batch, block, wave
    t0 = (head & 0xFFFF).astype(np.int64)
    return key, t0, masks


def popcount(a):
    a = a.astype(np.uint64)
    c = np.zeros(a.shape, np.int64)
    for _ in range(32):
        c += (a & 1).astype(np.int64)
        a >>= np.uint64(1)
    return c


def rounds_of(key, t0, counts):
    """lists of per-tile count arrays (64 lanes), one list per (wave, round)"""
    by_wave = defaultdict(list)
    for k, t, c in zip(key, t0, counts):
        by_wave[int(k)].append((int(t), c))
    out = []
    for tiles in by_wave.values():
        cur = []
        for t, c in tiles:
            if t == 0 and cur:
                out.append(cur)
                cur = []
            cur.append(c)
        if cur:
            out.append(cur)
    return out


def per_tile(rounds):
    return sum(int(c.max()) for r in rounds for c in r)


def merged(rounds, k):
    """k consecutive tiles of a round drained as one list per lane"""
    it = 0
    for r in rounds:
        for i in range(0, len(r), k):
            it += int(sum(r[i:i + k]).max())
    return it


def per_round(rounds):
    return sum(int(sum(r).max()) for r in rounds)


def sliding_exact(rounds_masks, half_bits):
    """the same with the real bit positions: masks per tile, steps of `half_bits` splats"""
    it = 0
    steps = 32 // half_bits
    for r in rounds_masks:
        backlog = np.zeros(64, np.int64)
        for m in r:
            for h in range(steps):
                shift = 32 - half_bits * (h + 1)
                part = (m >> np.uint32(shift)) & np.uint32((1 << half_bits) - 1)
                new = popcount(part)
                need = int(backlog.max())
                it += need
                backlog = np.maximum(backlog + new - need, 0) if need > 0 else backlog + new
        it += int(backlog.max())
    return it


def main():
    key, t0, masks = load(sys.argv[1])
    counts = popcount(masks)
    rounds = rounds_of(key, t0, counts)
    rounds_m = rounds_of(key, t0, masks)
    hits = int(counts.sum())
    tiles = len(t0)
    print("%d tiles of %d (wave, round)s, %.2f candidates per lane and tile" % (tiles, len(rounds), hits / tiles / 64))
    base = per_tile(rounds)
    rows = [("tile by tile (the kernel)", base),
            ("two tiles as one list", merged(rounds, 2)),
            ("four tiles as one list", merged(rounds, 4)),
            ("a round as one list", per_round(rounds)),
            ("window of 16 old + 16 new", sliding_exact(rounds_m, 16)),
            ("window of 24 old + 8 new", sliding_exact(rounds_m, 8)),
            ("steps of a whole tile, lookahead of one", sliding_exact(rounds_m, 32))]
    for name, it in rows:
        print("  %-42s %9d iterations  %5.3f of the kernel's  lane utilisation %.3f" % (name, it, it / base, hits / 64 / it))


if __name__ == "__main__":
    main()
