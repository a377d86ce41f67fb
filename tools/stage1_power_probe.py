#!/usr/bin/env python3
"""Is svk_c3d2_stage1 bound by the clock the chip holds under load?  Same launch on real features and on all-zero
features (identical instruction stream; zero operands toggle almost nothing in the matrix pipe)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from speaker_verification_amd.engine import get_engine
    from speaker_verification_amd.model import seeded_model
    eng = get_engine(0)
    emb = seeded_model(1, n_labels=4).to(eng.device).eval().fused_inference()
    t1, t2 = emb.stage1_tables(), emb.stage2_tables()
    n = 1024
    g = torch.Generator(device=eng.device)
    g.manual_seed(0)
    crops = torch.randint(0, 200, (n, 20), device=eng.device, dtype=torch.int32, generator=g)
    res = {}
    tz = tuple(torch.zeros_like(t) if torch.is_tensor(t) else t for t in t1)   # all-zero weights, biases and slopes
    for name, feat, t1 in (("random", torch.randn((n, 297, 40), device=eng.device, generator=g) * 2 - 6, t1),
                           ("zero features", torch.zeros((n, 297, 40), device=eng.device), t1),
                           ("zero features AND zero weights", torch.zeros((n, 297, 40), device=eng.device), tz),
                           ("random again", torch.randn((n, 297, 40), device=eng.device, generator=g) * 2 - 6, t1)):
        for _ in range(20):
            y = eng.c3d2_stage1(feat, crops, t1)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            y = eng.c3d2_stage1(feat, crops, t1)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = float(np.median(ts))
        res[name] = {"ms": ms, "tflops_algorithmic": n * 2 * (12.4416 + 143.327232) / 1e3 / ms}
    # memory-latency probe: 128 cubes (features stay in L2) against 1 024 / 4 096 (Infinity Cache / HBM): us per cube
    for n2 in (128, 1024, 4096):
        feat = torch.randn((n2, 297, 40), device=eng.device, generator=g)
        cr = torch.randint(0, 200, (n2, 20), device=eng.device, dtype=torch.int32, generator=g)
        for _ in range(10):
            eng.c3d2_stage1(feat, cr, emb.stage1_tables())
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            eng.c3d2_stage1(feat, cr, emb.stage1_tables())
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res["us_per_cube_n%d" % n2] = float(np.median(ts)) * 1e3 / n2
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
