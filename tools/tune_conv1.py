#!/usr/bin/env python3
"""conv1_1 (1 -> 16 channels, kernel (3,1,5)) as an explicit patch matrix x weight GEMM vs MIOpen."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(torch, fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    import torch
    import torch.nn.functional as F
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda", 0)
    B = 978
    x = torch.randn(B, 1, 20, 80, 40, device=dev)
    w = torch.randn(16, 1, 3, 1, 5, device=dev) * 0.1
    b = torch.randn(16, device=dev)
    res = {}
    ref = F.conv3d(x, w.contiguous(memory_format=torch.channels_last_3d), b)
    res["conv3d_ms"] = timeit(torch, lambda: F.conv3d(x, w, b))
    xs = x[:, 0]                                               # (B, 20, 80, 40)
    sD, sH, sW = 80 * 40, 40, 1

    def patches():
        v = xs.as_strided((B, 18, 80, 36, 3, 5), (20 * 80 * 40, sD, sH, sW, sD, sW))
        return v.reshape(B * 18 * 80 * 36, 15)                 # one gather-copy kernel

    wm = w.reshape(16, 15).t().contiguous()                    # (15, 16)
    res["patches_ms"] = timeit(torch, patches)
    P = patches()
    res["addmm_ms"] = timeit(torch, lambda: torch.addmm(b, P, wm))
    res["total_gemm_path_ms"] = timeit(torch, lambda: torch.addmm(b, patches(), wm))
    out = torch.addmm(b, patches(), wm).view(B, 18, 80, 36, 16).permute(0, 4, 1, 2, 3)
    res["max_abs_diff"] = float((out - ref).abs().max())
    res["is_channels_last"] = bool(out.is_contiguous(memory_format=torch.channels_last_3d))
    # K padded to 16 (one extra zero-weight tap that re-reads a valid sample)
    wm16 = torch.cat([wm, torch.zeros(1, 16, device=dev)], 0).contiguous()

    def patches16():
        v = xs.as_strided((B, 18, 80, 36, 3, 5), (20 * 80 * 40, sD, sH, sW, sD, sW)).reshape(B * 18 * 80 * 36, 15)
        return F.pad(v, (0, 1))
    res["total_gemm_k16_ms"] = timeit(torch, lambda: torch.addmm(b, patches16(), wm16))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
