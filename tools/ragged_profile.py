#!/usr/bin/env python3
"""Per-batch stage times of the realistic-length workload (bench.py's ragged corpus: 2 048 clips of 4 .. 145 s through
embed_ragged_resident): clips, audio seconds and HIP-event ms of VAD / front end / CMVN / crops / network per batch."""
import os
import sys

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
slots = (lens + 7) // 8 * 8
offs = np.concatenate([[0], np.cumsum(slots)[:-1]]).astype(np.int64)
base, _ = synth.corpus_device(1024, dev, first_clip=0, utts_per_speaker=123)
flat = base.reshape(-1)
buf = torch.zeros((int(slots.sum()),), dtype=torch.int16, device=dev)
rng = np.random.default_rng(11)
for k in range(n):
    start = int(rng.integers(0, 1024 - 49)) * synth.CLIP_SAMPLES
    buf[offs[k]:offs[k] + lens[k]] = flat[start:start + int(lens[k])]
pipe = VerificationPipeline(seeded_model(2024, n_labels=1211), use_vad=True, normalize=True, preemph_cof=0.98, crop_rng="device",
                            micro_batch=1024)
pipe.embed_ragged_resident(buf, offs, lens)
torch.cuda.synchronize()
spans = []
pipe.embed_ragged_resident(buf, offs, lens, spans=spans)
torch.cuda.synchronize()
batches = list(pipe._ragged_batches(lens, 64 * 1024 * 1024))
net = [(name, a, e) for name, a, e in spans if name == "network"]
front = [(name, a, e) for name, a, e in spans if name != "network"]
per = len(front) // len(batches)
print("network passes (deferred, full micro-batches): " + "  ".join("%.3f ms" % a.elapsed_time(e) for _, a, e in net))
for b, (idx, total) in enumerate(batches):
    row = {name: a.elapsed_time(e) for name, a, e in front[b * per:(b + 1) * per]}
    ll = lens[idx]
    print("batch %d: %4d clips, %6.1f .. %6.1f s (%7.0f s of audio)  " % (b, len(idx), ll.min() / 16000, ll.max() / 16000, ll.sum() / 16000)
          + "  ".join("%s %.3f" % kv for kv in row.items()))
