#!/usr/bin/env python3
"""Element-wise / pooling formulations of the C3D2 forward at the real activation shapes."""
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
    dev = torch.device("cuda", 0)
    B = 978
    res = {}
    slope = torch.tensor([0.25], device=dev)
    for name, shape in (("act1_1", (B, 16, 18, 80, 36)), ("act1_2", (B, 16, 16, 36, 36)), ("act2_2", (B, 32, 12, 15, 15))):
        x = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last_3d)
        r = {}
        r["prelu"] = timeit(torch, lambda: F.prelu(x, slope))
        r["leaky_relu"] = timeit(torch, lambda: F.leaky_relu(x, 0.25))
        y = x.clone()
        r["leaky_relu_"] = timeit(torch, lambda: F.leaky_relu_(y, 0.25))
        r["where"] = timeit(torch, lambda: torch.where(x >= 0, x, x * 0.25))
        w2 = shape[-1] // 2 * 2                      # MaxPool3d floors: an odd last column is dropped
        ev, od = x[..., 0:w2:2], x[..., 1:w2:2]
        if name != "act1_1":
            r["max_pool3d"] = timeit(torch, lambda: F.max_pool3d(x, (1, 1, 2), (1, 1, 2)))
            r["maximum_slices"] = timeit(torch, lambda: torch.maximum(ev, od))
            r["pool_then_leaky_"] = timeit(torch, lambda: F.leaky_relu_(torch.maximum(ev, od), 0.25))
            r["prelu_then_pool"] = timeit(torch, lambda: F.max_pool3d(F.prelu(x, slope), (1, 1, 2), (1, 1, 2)))
            a = F.max_pool3d(F.prelu(x, slope), (1, 1, 2), (1, 1, 2))
            b = F.leaky_relu_(torch.maximum(ev, od), 0.25)
            r["same"] = bool(torch.equal(a, b))
            r["out_is_channels_last"] = bool(b.is_contiguous(memory_format=torch.channels_last_3d))
        res[name] = r
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
