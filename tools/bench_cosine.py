#!/usr/bin/env python3
"""All-pairs cosine kernel at the VoxCeleb1 verification shape and at the dev-set stress shape."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    if os.environ.get("SVK_TOOL_LIB"):      # tuning only: time another build of the library (A/B in one GPU call)
        from speaker_verification_amd import _lib
        _lib.LIB_PATH = os.environ["SVK_TOOL_LIB"]
    from speaker_verification_amd.engine import get_engine
    eng = get_engine(0)
    res = {}
    shapes = {"verif_4874x40": (4874, 40, 128), "dev_148642x1211": (148642, 1211, 128),
              "square_16384": (16384, 16384, 128)}
    if os.environ.get("SVK_COS_SHAPES"):    # e.g. "148642x1216,148480x1211"
        shapes = {sh: tuple(int(v) for v in sh.split("x")) + (128,) for sh in os.environ["SVK_COS_SHAPES"].split(",")}
    for name, (nt, ns, d) in shapes.items():
        t = torch.randn(nt, d, device=eng.device)
        e = torch.randn(ns, d, device=eng.device)
        for _ in range(3):
            out = eng.cosine_scores(t, e)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = eng.cosine_scores(t, e)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = float(np.median(ts))
        flop = 2.0 * nt * ns * d
        byts = (nt + ns) * d * 4 + nt * ns * 4
        ref = torch.nn.functional.normalize(t[:64]) @ torch.nn.functional.normalize(e[:64]).T
        res[name] = {"ms": ms, "tflops": flop / ms / 1e9, "frac_of_157TF_f32_mfma": flop / ms / 1e9 / 157.3,
                     "GBps": byts / ms / 1e6, "max_abs_diff": float((out[:64, :64] - ref).abs().max())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
