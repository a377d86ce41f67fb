#!/usr/bin/env python3
"""HIP-event times of the first block's kernel forms (svk_c3d2_stage1: direct, depth-transformed per fragment read "w",
t-plane form "t") on N cubes, interleaved A/B on one box.   python tools/time_stage1.py [n_cubes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_verification_amd import _lib                                    # noqa: E402
if os.environ.get("SVK_TOOL_LIB"):                                            # an experiment build (make stamps EXP=... TAG=...)
    _lib.LIB_PATH = os.environ["SVK_TOOL_LIB"]
from speaker_verification_amd.engine import get_engine                       # noqa: E402
from speaker_verification_amd.model import perturb_inference_state, seeded_model   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4018
eng = get_engine(0)
model = seeded_model(1, 8)
model.load_state_dict(perturb_inference_state(model.state_dict(), 2))
emb = model.to(eng.device).eval().fused_inference(channels_last=True)
t1 = emb.stage1_tables()
T = 297
feat = torch.randn((n, T, 40), device=eng.device) * 2 - 6
crops = torch.randint(0, T - 80, (n, 20), device=eng.device, dtype=torch.int32)
forms = {"w (transform per fragment read)": dict(depth_transform=True),
         "m (w + merged remainder tiles)": dict(depth_transform=True, merged_tiles=True), "t (t planes)": dict(t_planes=True)}
if "--direct" in sys.argv:
    forms["direct"] = dict()
outs = {k: eng.c3d2_stage1(feat, crops, t1, folded=False, **kw) for k, kw in forms.items()}
ref = outs["w (transform per fragment read)"]
for k, o in outs.items():
    print("%-34s max |diff| vs w / scale: %.2e" % (k, float((o - ref).abs().max()) / float(ref.abs().max())))
del outs
times = {k: [] for k in forms}
for rep in range(12):
    for k, kw in forms.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.c3d2_stage1(feat, crops, t1, folded=False, **kw)
        b.record()
        torch.cuda.synchronize()
        if rep >= 2:
            times[k].append(a.elapsed_time(b))
mfma = {"w (transform per fragment read)": 118080, "m (w + merged remainder tiles)": 36 * (400 + 18 * 144),
        "t (t planes)": 72 * (240 + 9 * 144), "direct": 36 * 4288}
for k, v in times.items():
    ms = float(np.median(v))
    print("%-34s %8.3f ms per %d cubes   issued %.3f of the f32 pipe (157.3 TFLOP/s), %d MFMA per cube"
          % (k, ms, n, n * mfma[k] * 2048 / ms / 1e9 / 157.3, mfma[k]))
