#!/usr/bin/env python3
"""HIP-event time of the second block (svk_c3d2_stage2 = conv2_1 + conv2_2 kernels) on N cubes, a checksum of its output
(to compare experiment builds, SVK_TOOL_LIB=...) and the same for conv3_1.   python tools/time_stage2.py [n_cubes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_verification_amd import _lib                                    # noqa: E402
if os.environ.get("SVK_TOOL_LIB"):
    _lib.LIB_PATH = os.environ["SVK_TOOL_LIB"]
from speaker_verification_amd.engine import get_engine                       # noqa: E402
from speaker_verification_amd.model import perturb_inference_state, seeded_model   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4018
eng = get_engine(0)
model = seeded_model(1, 8)
model.load_state_dict(perturb_inference_state(model.state_dict(), 2))
emb = model.to(eng.device).eval().fused_inference(channels_last=True)
t2, t31 = emb.stage2_tables(), emb.conv31_tables()
g = torch.Generator(device=eng.device)
g.manual_seed(0)
x = torch.randn((n, 16, 36, 18, 16), device=eng.device, generator=g)
y = eng.c3d2_stage2(x, t2, depth_transform=True)
z = eng.c3d2_conv31(y, t31, chunked=True)
print("stage2 checksum %.9e %.9e   conv3_1 checksum %.9e %.9e" % (float(y.double().sum()), float(y.double().abs().sum()),
                                                                   float(z.double().sum()), float(z.double().abs().sum())))


def med(fn, reps=12, warm=3):
    for _ in range(warm):
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


ms2 = med(lambda: eng.c3d2_stage2(x, t2, depth_transform=True))
ms31 = med(lambda: eng.c3d2_conv31(y, t31, chunked=True))
print("stage2 %.3f ms per %d cubes (issued %.3f of the pipe)   conv3_1 %.3f ms (issued %.3f)" %
      (ms2, n, n * 75264 * 2048 / ms2 / 1e9 / 157.3, ms31, n * 9600 * 2048 / ms31 / 1e9 / 157.3))
