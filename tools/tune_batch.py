#!/usr/bin/env python3
"""Batch-size / layout sweep of the fused C3D2 embedder (MIOpen find mode on)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from speaker_verification_amd.model import seeded_model
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda", 0)
    model = seeded_model(1).to(dev)
    res = []
    for cl in (True, False):
        emb = model.fused_inference(channels_last=cl)
        for B in [int(b) for b in sys.argv[1:]] or [64, 256, 1024]:
            x = torch.randn(B, 1, 20, 80, 40, device=dev)
            t0 = time.perf_counter()
            emb(x)
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                emb(x)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            ms = float(np.median(ts))
            res.append({"channels_last": cl, "batch": B, "ms": ms, "utt_per_s": B / ms * 1e3, "first_call_s": first})
            print(json.dumps(res[-1]), flush=True)


if __name__ == "__main__":
    main()
