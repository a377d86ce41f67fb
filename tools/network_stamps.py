#!/usr/bin/env python3
"""In-kernel phase stamps (s_memtime cycles per item and wave, and the clock the chip holds) of the network kernels, from a
-DSVK_TUNING build (make -C speaker_verification_amd/csrc stamps -> build_variants/libsvk_stamps.so; the shipped library has
none; SVK_TOOL_LIB picks another variant).  Each kernel first runs ~1 s back to back so that the stamped launch sees the
clock the chip HOLDS under that load.          python tools/network_stamps.py [n_cubes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                  # noqa: E402
from speaker_verification_amd import _lib                                    # noqa: E402
_lib.LIB_PATH = os.environ.get("SVK_TOOL_LIB", os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "build_variants", "libsvk_stamps.so"))
from speaker_verification_amd.engine import get_engine                       # noqa: E402
from speaker_verification_amd.model import seeded_model                      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4018
eng = get_engine(0)
emb = seeded_model(1, n_labels=4).to(eng.device).eval().fused_inference()
g = torch.Generator(device=eng.device)
g.manual_seed(0)
feat = torch.randn((n, 297, 40), device=eng.device, generator=g) * 2 - 6
crops = torch.randint(0, 200, (n, 20), device=eng.device, dtype=torch.int32, generator=g)
y = eng.c3d2_stage1(feat, crops, emb.stage1_tables())
runs = (
    ("svk_c3d2_stage1", lambda: eng.c3d2_stage1(feat, crops, emb.stage1_tables()), 200),
    ("svk_c3d2_stage2", lambda: eng.c3d2_stage2(y, emb.stage2_tables()), 200),
    ("svk_c3d2_conv32t", lambda x=torch.randn((n, 10, 8, 5, 15, 8), device=eng.device): eng.c3d2_conv32t(x, emb.conv32t_tables()), 100),
    ("svk_c3d2_conv41", lambda x=torch.randn((n, 8, 8, 45, 8), device=eng.device): eng.c3d2_conv41(x, emb.conv41_tables()), 100),
    ("svk_c3d2_conv42", lambda x=torch.randn((n, 6, 16, 27, 8), device=eng.device): eng.c3d2_conv42(x, emb.conv42_tables()), 100),
)
for name, fn, reps in runs:
    os.environ.pop("SVK_C3D2_STAMPS", None)
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print("---- %s: %.3f ms per %d cubes (stamped build)" % (name, a.elapsed_time(b) / reps, n), file=sys.stderr)
    os.environ["SVK_C3D2_STAMPS"] = "1"
    fn()
    torch.cuda.synchronize()
