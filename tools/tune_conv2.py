#!/usr/bin/env python3
"""conv1_2 (16 -> 16 channels, kernel (3,9,1), stride (1,2,1), N = 16 is a narrow implicit GEMM):
does widening N with a Toeplitz expansion along D or H (G output positions per 'super channel') pay,
permute back to NDHWC included?"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(torch, fn, reps=5):
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
    cl = torch.channels_last_3d
    x = torch.randn(B, 16, 18, 80, 36, device=dev).contiguous(memory_format=cl)
    w = (torch.randn(16, 16, 3, 9, 1, device=dev) * 0.05).contiguous(memory_format=cl)
    b = torch.randn(16, device=dev)
    ref = F.conv3d(x, w, b, stride=(1, 2, 1))
    res = {"conv3d_ms": timeit(torch, lambda: F.conv3d(x, w, b, stride=(1, 2, 1)))}
    # Toeplitz along D: G output depths per super-channel
    for G in (2, 4):
        wt = torch.zeros(G * 16, 16, 3 + G - 1, 9, 1, device=dev)
        for g in range(G):
            wt[g * 16:(g + 1) * 16, :, g:g + 3] = w
        wt = wt.contiguous(memory_format=cl)
        bt = b.repeat(G)
        od = 16

        def run():
            y = F.conv3d(x, wt, bt, stride=(G, 2, 1))                       # (B, G*16, od/G, 36, 36)
            y = y.view(B, G, 16, od // G, 36, 36).permute(0, 2, 3, 1, 4, 5).reshape(B, 16, od, 36, 36)
            return y.contiguous(memory_format=cl)
        out = run()
        res[f"D{G}"] = {"total_ms": timeit(torch, run),
                        "conv_only_ms": timeit(torch, lambda: F.conv3d(x, wt, bt, stride=(G, 2, 1))),
                        "max_abs_diff": float((out - ref).abs().max())}
    # Toeplitz along H (stride 2): G output rows per super-channel
    for G in (2, 3):
        kh = 9 + 2 * (G - 1)
        wt = torch.zeros(G * 16, 16, 3, kh, 1, device=dev)
        for g in range(G):
            wt[g * 16:(g + 1) * 16, :, :, 2 * g:2 * g + 9] = w
        wt = wt.contiguous(memory_format=cl)
        bt = b.repeat(G)

        def run():
            y = F.conv3d(x, wt, bt, stride=(1, 2 * G, 1))                   # (B, G*16, 16, 36/G, 36)
            y = y.view(B, G, 16, 16, 36 // G, 36).permute(0, 2, 3, 4, 1, 5).reshape(B, 16, 16, 36, 36)
            return y.contiguous(memory_format=cl)
        out = run()
        res[f"H{G}"] = {"total_ms": timeit(torch, run),
                        "conv_only_ms": timeit(torch, lambda: F.conv3d(x, wt, bt, stride=(1, 2 * G, 1))),
                        "max_abs_diff": float((out - ref).abs().max())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
