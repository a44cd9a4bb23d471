#!/usr/bin/env python3
"""Does the Infinity Cache serve a read that follows a write of the same buffer?  (The 2-D path's intermediate: written by the row
pass, read by the column pass.)  Times a streaming read (sum) of a buffer (a) right after it was written, (b) after 1 GiB of
other traffic, for several sizes; HIP events, median of 7."""
import torch

dev = "cuda"
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB of other traffic


def timed(fn):
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(e0, e1)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[3]


for mb in (16, 32, 64, 128, 192, 384):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, dtype=torch.float32, device=dev)

    def warm(e0, e1):
        x.fill_(1.0)
        e0.record()
        x.sum()
        e1.record()

    def cold(e0, e1):
        x.fill_(1.0)
        big.fill_(2.0)
        e0.record()
        x.sum()
        e1.record()

    tw, tc = timed(warm), timed(cold)
    print(f"{mb:4d} MB: read right after the write {tw * 1e3:8.1f} us = {mb * 1.048576 / tw:7.1f} GB/s | after 1 GiB of other writes {tc * 1e3:8.1f} us = {mb * 1.048576 / tc:7.1f} GB/s", flush=True)
