"""Time the tiled cosine kernel against the number of 128-row blocks (shows how many workgroups are
resident at once: the time steps up each time a new round of workgroups starts)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speaker_verification_amd.engine import get_engine

eng = get_engine(0)
ns, d = 1216, 128
e = torch.randn(ns, d, device=eng.device)
for blocks in (128, 256, 384, 512, 640, 768, 896, 1024, 1280, 1536, 2048):
    t = torch.randn(128 * blocks, d, device=eng.device)
    for _ in range(3):
        eng.cosine_scores(t, e)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.cosine_scores(t, e); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    print(f"blocks={blocks:5d}  {ms*1e3:8.1f} us  {2.0*128*blocks*ns*d/ms/1e9:6.1f} TFLOP/s", flush=True)
