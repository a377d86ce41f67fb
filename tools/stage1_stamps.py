#!/usr/bin/env python3
"""Phase timing of svk_c3d2_stage1 from in-kernel s_memtime stamps (a `-DSVK_TUNING` build of csrc/c3d2.hip:
build_variants/libsvk_stamps.so; the shipped library has no stamps).  Prints cycles per item and wave."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from speaker_verification_amd import _lib
    _lib.LIB_PATH = os.environ.get("SVK_TOOL_LIB", os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "build_variants",
                                                                 "libsvk_stamps.so"))
    from speaker_verification_amd.engine import get_engine
    from speaker_verification_amd.model import seeded_model
    eng = get_engine(0)
    emb = seeded_model(1, n_labels=4).to(eng.device).eval().fused_inference(channels_last=True)
    t1 = emb.stage1_tables()
    n = 1024
    g = torch.Generator(device=eng.device)
    g.manual_seed(0)
    feat = torch.randn((n, 297, 40), device=eng.device, generator=g) * 2 - 6
    crops = torch.randint(0, 200, (n, 20), device=eng.device, dtype=torch.int32, generator=g)
    dt = os.environ.get("SVK_C3D2_DEPTH_TRANSFORM", "1") != "0"
    for mode in (False, True):
        for _ in range(20):
            eng.c3d2_stage1(feat, crops, t1, folded=False, depth_transform=mode)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            eng.c3d2_stage1(feat, crops, t1, folded=False, depth_transform=mode)
        b.record()
        torch.cuda.synchronize()
        print("svk_c3d2_stage1, 1024 cubes, depth_transform=%s: %.3f ms" % (mode, a.elapsed_time(b) / 20), file=sys.stderr)
    t2 = emb.stage2_tables()
    y = eng.c3d2_stage1(feat, crops, t1, folded=False)
    for mode in (False, True):
        for _ in range(10):
            eng.c3d2_stage2(y, t2, depth_transform=mode)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            eng.c3d2_stage2(y, t2, depth_transform=mode)
        b.record()
        torch.cuda.synchronize()
        print("svk_c3d2_stage2, 1024 cubes, depth_transform=%s: %.3f ms" % (mode, a.elapsed_time(b) / 20), file=sys.stderr)
    os.environ["SVK_C3D2_STAMPS"] = "1"
    y = eng.c3d2_stage1(feat, crops, t1, folded=False, depth_transform=dt)
    torch.cuda.synchronize()
    eng.c3d2_stage2(y, t2, depth_transform=dt)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
