"""Micro-batch sweep of the channels-last fused embedder (sizes that divide the 18 581-clip shard evenly)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speaker_verification_amd.model import seeded_model
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
emb = seeded_model(1).to(dev).fused_inference(channels_last=True)
for B in [int(b) for b in sys.argv[1:]] or [489, 581, 775, 978, 1162, 1549, 2323, 3097]:
    x = torch.randn(B, 1, 20, 80, 40, device=dev)
    for _ in range(2):
        emb(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); emb(x); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    print(f"batch {B:5d}  {ms:8.3f} ms  {B / ms * 1e3:9.0f} cubes/s", flush=True)
    del x
    torch.cuda.empty_cache()
