#!/usr/bin/env python3
"""HIP-event time, issued share of the f32 matrix pipe and an output checksum of every network kernel (svk_c3d2_stage1,
stage2, conv31, conv32t, conv41, conv42, fc5) on N cubes, each fed random activations of its own input layout.  With
SVK_TOOL_LIB=<experiment build> (make -C speaker_verification_amd/csrc stamps EXP=-D... TAG=_name) the checksums tell
whether the experiment changed results.      python tools/time_network.py [n_cubes] [kernel ...]"""
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

args = [a for a in sys.argv[1:]]
n = int(args.pop(0)) if args and args[0].isdigit() else 4018
eng = get_engine(0)
model = seeded_model(1, 8)
model.load_state_dict(perturb_inference_state(model.state_dict(), 2))
emb = model.to(eng.device).eval().fused_inference()
g = torch.Generator(device=eng.device)
g.manual_seed(0)


def rnd(*shape):
    return torch.randn(shape, device=eng.device, generator=g)


T = 297
feat = rnd(n, T, 40) * 2 - 6
crops = torch.randint(0, T - 80, (n, 20), device=eng.device, dtype=torch.int32, generator=g)
# name: (call, MFMA wave-instructions per cube by construction, direct-form multiply-adds per cube)
kernels = {
    # the first block runs two-piece f16 products: its MFMAs (v_mfma_f32_16x16x32_f16, 16 cycles) counted in units of the f32 MFMA's
    # 32 cycles -- 36 items x (100 conv1_1 tiles x 2 + 36 conv1_2 tiles x 42) / 2 -- so that "issued" stays a share of issue TIME
    "stage1": (lambda: eng.c3d2_stage1(feat, crops, emb.stage1_tables()), 36 * (100 * 2 + 36 * 42) / 2, 155.768832e6),
    "stage2": (lambda x=rnd(n, 16, 36, 18, 16): eng.c3d2_stage2(x, emb.stage2_tables()), (9 * 49 * 36 + 21 * 8 * 2 * 72) / 2, 112.80384e6),
    "conv3_1": (lambda x=rnd(n, 12, 15, 7, 32): eng.c3d2_conv31(x, emb.conv31_tables()), 5 * 10 * 4 * 27 / 2, 13.824e6),
    "conv3_2": (lambda x=rnd(n, 10, 8, 5, 15, 8): eng.c3d2_conv32t(x, emb.conv32t_tables()), 5 * 5 * 4 * 126 / 2, 30.96576e6),
    "conv4_1": (lambda x=rnd(n, 8, 8, 45, 8): eng.c3d2_conv41(x, emb.conv41_tables()), 11 * 8 * 54 / 2, 11.943936e6),
    "conv4_2": (lambda x=rnd(n, 6, 16, 27, 8): eng.c3d2_conv42(x, emb.conv42_tables()), 8064, 12.386304e6),
    "fc5": (lambda x=rnd(n, 4, 16, 9, 8): eng.c3d2_fc5(x, emb.fc5_tables()), 576, 0.589824e6),
}


def med(fn, reps=20, warm=5):
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


total = 0.0
for name, (fn, mfma, mmac) in kernels.items():
    if args and name not in args:
        continue
    out = fn()
    check = float(out.double().abs().sum())
    ms = med(fn)
    total += ms
    print("%-8s %8.3f ms per %d cubes   issued %.3f of its matrix pipe's time   direct-form %.3f   checksum %.9e"
          % (name, ms, n, n * mfma * 2048 / ms / 1e9 / 157.3, 2 * mmac * n / ms / 1e9 / 157.3, check))
print("total %.3f ms per %d cubes = %.0f cubes/s" % (total, n, n / total * 1e3))
