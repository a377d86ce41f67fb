"""Front-end kernel time against the number of clips per launch (fixed launch cost vs per-clip cost).
Usage (GPU box): python tools/fe_scaling.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speaker_verification_amd import _lib, synth
if os.environ.get("SVK_TOOL_LIB"):      # tuning only: time another build of the library (A/B in one GPU call)
    _lib.LIB_PATH = os.environ["SVK_TOOL_LIB"]
from speaker_verification_amd.engine import get_engine, spec_from_seconds

eng = get_engine(0)
specs = {"A": spec_from_seconds(16000, 0.020, 0.01, 512, 40, 13, _lib.OUT_MFCC, preemph=True, preemph_cof=0.98),
         "B": spec_from_seconds(16000, 0.025, 0.01, 1024, 40, 40, _lib.OUT_LMFE, preemph=True, preemph_cof=0.98)}
base = np.stack([synth.noise_clip(s) for s in range(16)])
for name, spec in specs.items():
    for n in [int(v) for v in os.environ.get("SVK_TOOL_SIZES", "64,256,768,1024,2048,4096,8192").split(",")]:
        pcm = eng.to_device(np.tile(base, (n // 16, 1)))
        for _ in range(3):
            eng.features(pcm, spec)
        torch.cuda.synchronize()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(12)]
        for a, b in ev:
            a.record()
            eng.features(pcm, spec)
            b.record()
        torch.cuda.synchronize()
        t = float(np.median([a.elapsed_time(b) for a, b in ev]))
        print(f"{name} n={n:5d}  {t * 1e3:8.1f} us  {t * 1e3 / n:7.4f} us/clip", flush=True)
