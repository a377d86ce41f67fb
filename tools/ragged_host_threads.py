#!/usr/bin/env python3
"""The host-fed realistic-length workload (bench.py's ragged corpus through VerificationPipeline.embed_ragged) against the
number of host packing threads (SVK_RAGGED_THREADS): median of five runs each.   python tools/ragged_host_threads.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                   # noqa: E402
from speaker_verification_amd import synth                                    # noqa: E402
from speaker_verification_amd.engine import get_engine                        # noqa: E402
from speaker_verification_amd.model import seeded_model                       # noqa: E402
from speaker_verification_amd.pipeline import VerificationPipeline            # noqa: E402

eng = get_engine(0)
dev = eng.device
n = 2048
lens = bench.ragged_lengths(n)
base, _ = synth.corpus_device(1024, dev, first_clip=0, utts_per_speaker=123)
flat = base.reshape(-1).cpu().numpy()
rng = np.random.default_rng(11)
clips = []
for k in range(n):
    start = int(rng.integers(0, 1024 - 49)) * synth.CLIP_SAMPLES
    clips.append(flat[start:start + int(lens[k])].copy())
pipe = VerificationPipeline(seeded_model(2024, n_labels=1211), use_vad=True, normalize=True, preemph_cof=0.98, crop_rng="device",
                            micro_batch=1024)
for threads in (4, 8, 12, 16, 24, 32):
    os.environ["SVK_RAGGED_THREADS"] = str(threads)
    pipe._rag_cap = 0                      # rebuild the staging buffers and the pool
    pipe.embed_ragged(clips)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        pipe.embed_ragged(clips)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("%2d packing threads: median %.1f ms (min %.1f, max %.1f) = %.1f k utt/s" %
          (threads, 1e3 * np.median(ts), 1e3 * min(ts), 1e3 * max(ts), n / np.median(ts) / 1e3), flush=True)
