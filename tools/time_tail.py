#!/usr/bin/env python3
"""HIP-event times of the last block's kernels (svk_c3d2_conv41 / conv42 / fc5, csrc/c3d2_tail.hip) on N cubes, beside
what they replaced (MIOpen's conv4_1 / conv4_2 + svk_bias_prelu + hipBLASLt FC5 through FusedEmbedder._run).
    python tools/time_tail.py [n_cubes]"""
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
t41, t42, tfc = emb.conv41_tables(), emb.conv42_tables(), emb.fc5_tables()
x1 = torch.randn((n, 8, 8, 45, 8), device=eng.device)
x2 = torch.randn((n, 6, 16, 27, 8), device=eng.device)
x3 = torch.randn((n, 4, 16, 9, 8), device=eng.device)


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


mac = {"conv4_1": 11.943936e6, "conv4_2": 12.386304e6, "fc5": 0.589824e6}
rows = {"conv4_1": med(lambda: eng.c3d2_conv41(x1, t41)), "conv4_2": med(lambda: eng.c3d2_conv42(x2, t42)),
        "fc5": med(lambda: eng.c3d2_fc5(x3, tfc))}
for k, ms in rows.items():
    tf = 2 * mac[k] * n / ms / 1e9
    issued = tf * (2.0 / 3.0 if k != "fc5" else 1.0)
    print("%-8s %8.3f ms  %6.1f TFLOP/s direct-form (%.3f of 157.3), issued %.3f of the pipe" % (k, ms, tf, tf / 157.3, issued / 157.3))
print("tail total %.3f ms per %d cubes" % (sum(rows.values()), n))
if os.environ.get("SVK_TOOL_LIB"):
    sys.exit(0)
# what it replaces: MIOpen convolutions (exhaustive find) + svk_bias_prelu + F.linear
xin = torch.randn((n, 8, 9, 5, 64), device=eng.device).permute(0, 4, 1, 2, 3)
saved = torch.backends.cudnn.benchmark
torch.backends.cudnn.benchmark = True
try:
    old = med(lambda: emb._run(xin, start=6), reps=10, warm=3)
finally:
    torch.backends.cudnn.benchmark = saved
print("PyTorch-ROCm tail (MIOpen conv4_1, conv4_2 + svk_bias_prelu x 2 + F.linear): %.3f ms" % old)
