#!/usr/bin/env python3
"""Does the chip hold its clock under the network kernels for longer than a second?  Launches the first block (then the
second) back to back on 4 018 cubes for ~12 s each; prints a launch's HIP-event time every so often and, from the
-DSVK_TUNING build (make -C speaker_verification_amd/csrc stamps), the in-kernel clock (s_memtime cycles per
s_memrealtime tick) of a stamped launch at the same points.   python tools/clock_under_load.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                  # noqa: E402
from speaker_verification_amd import _lib                                    # noqa: E402
_lib.LIB_PATH = os.environ.get("SVK_TOOL_LIB", os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "build_variants", "libsvk_stamps.so"))
from speaker_verification_amd.engine import get_engine                       # noqa: E402
from speaker_verification_amd.model import seeded_model                      # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
eng = get_engine(0)
emb = seeded_model(1, n_labels=4).to(eng.device).eval().fused_inference()
t1, t2 = emb.stage1_tables(), emb.stage2_tables()
n = 4018
g = torch.Generator(device=eng.device)
g.manual_seed(0)
feat = torch.randn((n, 297, 40), device=eng.device, generator=g) * 2 - 6
crops = torch.randint(0, 200, (n, 20), device=eng.device, dtype=torch.int32, generator=g)
y = eng.c3d2_stage1(feat, crops, t1)


def run(name, fn):
    torch.cuda.synchronize()
    time.sleep(2.0)                      # start from an idle chip
    t0 = time.time()
    k = 0
    while time.time() - t0 < secs:
        for _ in range(40):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        print("%s  t = %5.2f s  launch %.3f ms" % (name, time.time() - t0, a.elapsed_time(b)), file=sys.stderr, flush=True)
        k += 1
        if k % 4 == 1:
            os.environ["SVK_C3D2_STAMPS"] = "1"
            fn()
            torch.cuda.synchronize()
            os.environ.pop("SVK_C3D2_STAMPS", None)


run("stage1", lambda: eng.c3d2_stage1(feat, crops, t1))
run("stage2", lambda: eng.c3d2_stage2(y, t2))
