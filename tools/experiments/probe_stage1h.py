#!/usr/bin/env python3
"""svk_c3d2_stage1h against svk_c3d2_stage1 with conv1_2 reduced to ONE tap (identity over the channels), tap by tap: which
tap pair / which half of a K = 32 block / which piece goes wrong.      python tools/experiments/probe_stage1h.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speaker_verification_amd.engine import get_engine                       # noqa: E402
from speaker_verification_amd.model import seeded_model                      # noqa: E402

eng = get_engine(0)
dev = eng.device
g = torch.Generator(device=dev)
g.manual_seed(0)
n, T = 8, 297
feat = torch.randn((n, T, 40), device=dev, generator=g) * 2 - 1
crops = torch.randint(0, T - 80, (n, 20), device=dev, dtype=torch.int32, generator=g)


def run(tag, edit):
    m = seeded_model(1, 8)
    with torch.no_grad():
        for bn in (m.batch_norm1_1, m.batch_norm1_2):
            bn.running_mean.zero_()
            bn.running_var.fill_(1.0)
            bn.weight.fill_(1.0)
            bn.bias.zero_()
        edit(m)
    emb = m.to(dev).eval().fused_inference()
    a = eng.c3d2_stage1(feat, crops, emb.stage1_tables())
    b = eng.c3d2_stage1h(feat, crops, emb.stage1h_tables())
    d = (a - b).abs()
    bad = d > 1e-4 * a.abs().max()
    where = ""
    if bad.any():
        idx = bad.nonzero()
        where = "  bad %d of %d; first %s; depths %s rows %s cols %s ch %s" % (
            int(bad.sum()), bad.numel(), idx[0].tolist(), sorted(set(idx[:, 1].tolist()))[:8], sorted(set(idx[:, 2].tolist()))[:8],
            sorted(set(idx[:, 3].tolist()))[:8], sorted(set(idx[:, 4].tolist()))[:16])
    print("%-28s max|d| %.3e  scale %.3e%s" % (tag, float(d.max()), float(a.abs().max()), where), flush=True)


def one_tap(kd, kh, w11=None):
    def edit(m):
        m.conv1_2.weight.zero_()
        m.conv1_2.bias.zero_()
        for c in range(16):
            m.conv1_2.weight[c, c, kd, kh, 0] = 1.0
        m.PReLu1_2.weight.fill_(1.0)
        if w11 is not None:
            w11(m)
    return edit


def single_conv11_tap(kd, kw):
    def f(m):
        m.conv1_1.weight.zero_()
        m.conv1_1.bias.zero_()
        for c in range(16):
            m.conv1_1.weight[c, 0, kd, 0, kw] = 1.0 + 0.125 * c
        m.PReLu1_1.weight.fill_(1.0)
    return f


for kd, kw in ((0, 0), (0, 4), (1, 2), (1, 3), (2, 4)):
    run("conv1_1 tap (%d,%d) only, conv1_2 (0,0)" % (kd, kw), one_tap(0, 0, single_conv11_tap(kd, kw)))
for kd, kh in ((0, 0), (0, 1), (0, 6), (0, 7), (0, 8), (1, 0), (1, 8), (2, 7), (2, 8)):
    run("conv1_2 tap (%d,%d)" % (kd, kh), one_tap(kd, kh))
run("both random", lambda m: None)
