#!/usr/bin/env python3
"""What the host link gives a bucket stream: host-to-device copies of one bucket's size (62 MB), alone, split over two
streams, and next to device-to-host read-backs of a ship-out's size -- the floor of SURVEY 8(d)'s region."""
import json
import time

import torch


def run(h2d_streams, with_d2h, with_fill, n=40, mb=62, d2h_mb=23):
    dev = torch.device("cuda", 0)
    src = [torch.empty(mb << 20, dtype=torch.uint8).pin_memory() for _ in range(4)]
    dst = [torch.empty(mb << 20, dtype=torch.uint8, device=dev) for _ in range(4)]
    back_src = torch.empty(d2h_mb << 20, dtype=torch.uint8, device=dev)
    back_dst = [torch.empty(d2h_mb << 20, dtype=torch.uint8).pin_memory() for _ in range(4)]
    page = torch.empty(mb << 20, dtype=torch.uint8)
    streams = [torch.cuda.Stream() for _ in range(h2d_streams)]
    back = torch.cuda.Stream()
    torch.cuda.synchronize()
    import threading
    stop = [False]

    def filler():
        k = 0
        while not stop[0]:
            src[(k + 2) % 4][: 16 << 20].copy_(page[: 16 << 20])     # a host thread writing pinned memory meanwhile
            k += 1
    th = threading.Thread(target=filler) if with_fill else None
    if th:
        th.start()
    t0 = time.perf_counter()
    part = (mb << 20) // h2d_streams
    for i in range(n):
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                dst[i % 4][k * part:(k + 1) * part].copy_(src[i % 4][k * part:(k + 1) * part], non_blocking=True)
        if with_d2h:
            with torch.cuda.stream(back):
                back_dst[i % 4].copy_(back_src, non_blocking=True)
    for s in streams:
        s.synchronize()
    t_h2d = time.perf_counter() - t0
    back.synchronize()
    t_all = time.perf_counter() - t0
    stop[0] = True
    if th:
        th.join()
    return {"h2d_streams": h2d_streams, "d2h": with_d2h, "fill": with_fill, "h2d_GBps": round(n * (mb << 20) / t_h2d / 1e9, 1),
            "d2h_GBps": round(n * (d2h_mb << 20) / t_all / 1e9, 1) if with_d2h else None}


if __name__ == "__main__":
    out = []
    for hs in (1, 2, 4):
        for d2h in (False, True):
            for fill in (False, True):
                run(hs, d2h, fill, n=8)
                out.append(run(hs, d2h, fill))
    for r in out:
        print(json.dumps(r))
