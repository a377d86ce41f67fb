#!/usr/bin/env python3
"""Per-batch HIP-event times of the non-network stages of the realistic-length workload (bench.py `ragged`: 2 048 clips of
4 .. 145 s in one resident buffer): which length-sorted batch costs what in the VAD, the front end, the CMVN statistics, the
crop draw and the cube gather, with the PCM bytes each batch streams.      python tools/time_ragged_front.py [n_clips]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                                                  # noqa: E402
from speaker_verification_amd import constants as c, synth                    # noqa: E402
from speaker_verification_amd.engine import get_engine                        # noqa: E402
from speaker_verification_amd.pipeline import VerificationPipeline            # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eng = get_engine(0)
dev = eng.device
from speaker_verification_amd.model import seeded_model                      # noqa: E402  (the front end does not read the weights)
pipe = VerificationPipeline(seeded_model(1, n_labels=4).to(dev).eval(), use_vad=True, normalize=True, preemph_cof=0.98,
                            crop_rng="device", micro_batch=1024)
lens = bench.ragged_lengths(n_clips)
slots = (lens + 7) // 8 * 8
offs = np.concatenate([[0], np.cumsum(slots)[:-1]]).astype(np.int64)
base, _ = synth.corpus_device(1024, dev, first_clip=0, utts_per_speaker=bench.UTTS_PER_SPK)
flat = base.reshape(-1)
buf = torch.zeros((int(slots.sum()),), dtype=torch.int16, device=dev)
rng = np.random.default_rng(11)
for k in range(n_clips):
    start = int(rng.integers(0, 1024 - 49)) * synth.CLIP_SAMPLES
    buf[offs[k]:offs[k] + lens[k]] = flat[start:start + int(lens[k])]
plan = pipe._ragged_batches(lens.astype(np.int32), 64 * 1024 * 1024)
rows = []
for rep in range(4):
    rows = []
    for batch, samples in plan:
        ids = np.asarray(batch, dtype=np.int64)
        o = torch.from_numpy(offs[ids]).to(dev)
        ln = torch.from_numpy(lens[ids].astype(np.int32)).to(dev)
        keys = torch.from_numpy(ids).to(dev)
        spans = []
        feat, idx, stats = pipe._ragged_front(buf, o, ln, int(lens[ids].max()), keys, spans=spans)
        torch.cuda.synchronize()
        by = {name: a.elapsed_time(b) for name, a, b in spans}
        rows.append((len(ids), float(lens[ids].min()) / 16000, float(lens[ids].max()) / 16000, 2 * samples, by))
tot = {}
for n, lo, hi, nbytes, by in rows:
    print("%5d clips of %6.1f .. %6.1f s  %7.1f MB   " % (n, lo, hi, nbytes / 1e6) +
          "  ".join("%s %.3f ms" % (k, v) for k, v in by.items()) + "   vad %.0f GB/s" % (nbytes / by["vad"] / 1e6))
    for k, v in by.items():
        tot[k] = tot.get(k, 0.0) + v
print("sum: " + "  ".join("%s %.3f ms" % (k, v) for k, v in tot.items()))
