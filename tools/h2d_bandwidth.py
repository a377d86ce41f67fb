#!/usr/bin/env python3
"""Host-to-device copy rate of a pinned / pageable 128 MB buffer on this box (what bounds the host-fed paths)."""
import time
import torch
n = 64 * 1024 * 1024
pinned = torch.empty((n,), dtype=torch.int16).pin_memory()
pageable = torch.empty((n,), dtype=torch.int16)
dev = torch.empty((n,), dtype=torch.int16, device="cuda:0")
for name, src in (("pinned", pinned), ("pageable", pageable)):
    for _ in range(2):
        dev.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        dev.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("%-8s 128 MB host -> device: median %.2f ms = %.1f GB/s" % (name, 1e3 * ts[3], n * 2 / ts[3] / 1e9))
