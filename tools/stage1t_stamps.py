#!/usr/bin/env python3
"""In-kernel phase stamps of the first block's t-plane form (c3d2_stage1t_kernel) and of the round-2 form beside it:
cycles per item and wave, from a -DSVK_TUNING build (make -C speaker_verification_amd/csrc stamps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                  # noqa: E402
from speaker_verification_amd import _lib                                    # noqa: E402
_lib.LIB_PATH = os.environ.get("SVK_TOOL_LIB", os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "build_variants", "libsvk_stamps.so"))
from speaker_verification_amd.engine import get_engine                       # noqa: E402
from speaker_verification_amd.model import seeded_model                      # noqa: E402

eng = get_engine(0)
emb = seeded_model(1, n_labels=4).to(eng.device).eval().fused_inference(channels_last=True)
t1 = emb.stage1_tables()
n = 1024
g = torch.Generator(device=eng.device)
g.manual_seed(0)
feat = torch.randn((n, 297, 40), device=eng.device, generator=g) * 2 - 6
crops = torch.randint(0, 200, (n, 20), device=eng.device, dtype=torch.int32, generator=g)
for kw in (dict(depth_transform=True), dict(depth_transform=True, merged_tiles=True), dict(t_planes=True)):
    for _ in range(10):
        eng.c3d2_stage1(feat, crops, t1, folded=False, **kw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        eng.c3d2_stage1(feat, crops, t1, folded=False, **kw)
    b.record()
    torch.cuda.synchronize()
    print("svk_c3d2_stage1 %s: %.3f ms per 1024 cubes (stamped build)" % (kw, a.elapsed_time(b) / 10), file=sys.stderr)
for kw in (dict(depth_transform=True), dict(depth_transform=True, merged_tiles=True), dict(t_planes=True)):
    print("---- %s" % kw, file=sys.stderr)
    os.environ.pop("SVK_C3D2_STAMPS", None)
    for _ in range(700):                       # ~1.5 s of back-to-back launches: the clock the chip HOLDS under this load
        eng.c3d2_stage1(feat, crops, t1, folded=False, **kw)
    os.environ["SVK_C3D2_STAMPS"] = "1"
    eng.c3d2_stage1(feat, crops, t1, folded=False, **kw)
    torch.cuda.synchronize()

# second block: conv2_2's phases (conv22w stamps)
t2 = emb.stage2_tables()
y = eng.c3d2_stage1(feat, crops, t1, folded=False, depth_transform=True, merged_tiles=True)
os.environ.pop("SVK_C3D2_STAMPS", None)
for _ in range(700):                           # ~1 s of back-to-back launches
    eng.c3d2_stage2(y, t2, depth_transform=True)
os.environ["SVK_C3D2_STAMPS"] = "1"
eng.c3d2_stage2(y, t2, depth_transform=True)
torch.cuda.synchronize()

# the last block's kernels (and conv3_2 in their shape): phases / barriers / epilogue per item, 4 018 cubes
n4 = 4018
os.environ.pop("SVK_C3D2_STAMPS", None)
x32 = torch.randn((n4, 10, 8, 5, 15, 8), device=eng.device)
x41 = torch.randn((n4, 8, 8, 45, 8), device=eng.device)
x42 = torch.randn((n4, 6, 16, 27, 8), device=eng.device)
for fn, x, t in ((eng.c3d2_conv32t, x32, emb.conv32t_tables()), (eng.c3d2_conv41, x41, emb.conv41_tables()), (eng.c3d2_conv42, x42, emb.conv42_tables())):
    os.environ.pop("SVK_C3D2_STAMPS", None)
    for _ in range(20):
        fn(x, t)
    os.environ["SVK_C3D2_STAMPS"] = "1"
    fn(x, t)
    torch.cuda.synchronize()
