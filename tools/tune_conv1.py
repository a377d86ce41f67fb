#!/usr/bin/env python3
"""conv1_1 (1 -> 16 channels, kernel (3,1,5)) as patch-matrix x weight GEMMs vs MIOpen.
Variant G groups G adjacent output columns into one GEMM row: the row reads a 3 x (5+G-1) window and
produces G x 16 outputs through a Toeplitz-expanded weight matrix (more FLOPs, far fewer bytes, and a
GEMM shape the library handles well); its row-major output is still NDHWC."""
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
    xs = x[:, 0]
    D, H, W = 20, 80, 40
    kd, kw, co = 3, 5, 16
    od, ow = D - kd + 1, W - kw + 1
    for G in (1, 2, 4, 6, 9, 12, 18, 36):
        win = kw + G - 1
        wt = torch.zeros(kd, win, G, co, device=dev)
        for g in range(G):
            wt[:, g:g + kw, g, :] = w[:, 0, :, 0, :].permute(1, 2, 0)
        wt = wt.reshape(kd * win, G * co).contiguous()
        bt = b.repeat(G)

        def run():
            p = xs.as_strided((B, od, H, ow // G, kd, win), (D * H * W, H * W, W, G, H * W, 1))
            return torch.addmm(bt, p.reshape(B * od * H * (ow // G), kd * win), wt)
        out = run().view(B, od, H, ow, co).permute(0, 4, 1, 2, 3)
        res[f"G{G}"] = {"ms": timeit(torch, run), "max_abs_diff": float((out - ref).abs().max()),
                        "gflop": 2.0 * B * od * H * (ow // G) * kd * win * G * co / 1e9}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
