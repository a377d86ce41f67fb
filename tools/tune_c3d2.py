#!/usr/bin/env python3
"""Per-layer timing of the C3D2 forward on PyTorch-ROCm under several formulations
(layout, MIOpen find mode, conv3d vs equivalent conv2d), to pick the fastest exact-f32 one."""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--benchmark", action="store_true")
    args = ap.parse_args()
    import torch
    import torch.nn.functional as F
    from speaker_verification_amd.model import _LAYERS, seeded_model
    torch.backends.cudnn.benchmark = args.benchmark
    dev = torch.device("cuda", 0)
    model = seeded_model(1).to(dev)
    B = args.batch
    x = torch.randn(B, 1, 20, 80, 40, device=dev)
    out = {"batch": B, "benchmark": args.benchmark, "layers": {}}
    with torch.no_grad():
        fused = model.fused_inference()
        out["fused_total_ms"] = timeit(torch, lambda: fused(x))
        fused_cl = model.fused_inference(channels_last=True)
        out["fused_channels_last_total_ms"] = timeit(torch, lambda: fused_cl(x))
        cur = x
        for (tag, _, cout, kernel, stride, pool), (w, b, slope, _, _, _) in zip(_LAYERS, fused.stages):
            rec = {"in": list(cur.shape), "kernel": list(kernel), "stride": list(stride)}
            rec["conv3d_ms"] = timeit(torch, lambda: F.conv3d(cur, w, b, stride=stride))
            y = F.conv3d(cur, w, b, stride=stride)
            rec["prelu_ms"] = timeit(torch, lambda: F.prelu(y, slope))
            cur_cl = cur.contiguous(memory_format=torch.channels_last_3d)
            w_cl = w.contiguous(memory_format=torch.channels_last_3d)
            rec["conv3d_cl_ms"] = timeit(torch, lambda: F.conv3d(cur_cl, w_cl, b, stride=stride))
            # equivalent conv2d: the kernel has extent 1 along H (k=(3,1,5)) or along W (k=(3,9,1))
            n, c, d, h, wd = cur.shape
            if kernel[1] == 1:      # fold H into the batch: (N*H, C, D, W)
                x2 = cur.permute(0, 3, 1, 2, 4).reshape(n * h, c, d, wd)
                w2 = w[:, :, :, 0, :]
                rec["conv2d_ms"] = timeit(torch, lambda: F.conv2d(x2, w2, b, stride=(stride[0], stride[2])))
                rec["conv2d_incl_permute_ms"] = timeit(
                    torch, lambda: F.conv2d(cur.permute(0, 3, 1, 2, 4).reshape(n * h, c, d, wd), w2, b,
                                            stride=(stride[0], stride[2])))
            else:                   # fold W into the batch: (N*W, C, D, H)
                x2 = cur.permute(0, 4, 1, 2, 3).reshape(n * wd, c, d, h)
                w2 = w[:, :, :, :, 0]
                rec["conv2d_ms"] = timeit(torch, lambda: F.conv2d(x2, w2, b, stride=(stride[0], stride[1])))
                rec["conv2d_incl_permute_ms"] = timeit(
                    torch, lambda: F.conv2d(cur.permute(0, 4, 1, 2, 3).reshape(n * wd, c, d, h), w2, b,
                                            stride=(stride[0], stride[1])))
            macs = cout * y.shape[2] * y.shape[3] * y.shape[4] * w.shape[1] * kernel[0] * kernel[1] * kernel[2]
            rec["gflop_per_sample"] = 2 * macs / 1e9
            rec["conv3d_tflops"] = 2 * macs * B / rec["conv3d_ms"] / 1e9
            cur = F.prelu(y, slope)
            if pool:
                rec["pool_ms"] = timeit(torch, lambda: F.max_pool3d(cur, (1, 1, 2), (1, 1, 2)))
                cur = F.max_pool3d(cur, (1, 1, 2), (1, 1, 2))
            out["layers"][tag] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
