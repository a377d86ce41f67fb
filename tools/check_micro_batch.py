#!/usr/bin/env python3
"""Embeddings of the WHOLE 148 642-clip corpus with micro-batches of 4 096 clips and of half the corpus (74 321 clips = 7.1e9 samples
behind one base pointer: every kernel's offsets must be 64-bit): equal bit for bit -- an embedding depends on its clip alone.  The
bench's EER only looks at the first 4 874 clips; this looks at all of them.      python tools/check_micro_batch.py [big] [small]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_verification_amd import synth                                   # noqa: E402
from speaker_verification_amd.model import C3D2                              # noqa: E402
from speaker_verification_amd.pipeline import VerificationPipeline           # noqa: E402

big = int(sys.argv[1]) if len(sys.argv) > 1 else 74321
small = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = 148642
dev = torch.device("cuda", 0)
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ck = torch.load(os.path.join(here, "speaker_verification_amd", "checkpoints", "c3d2_synth.pt"), map_location="cpu", weights_only=True)
model = C3D2(int(ck["state_dict"]["FC6.weight"].shape[0]), 1)
model.load_state_dict(ck["state_dict"])
pcm, _ = synth.corpus_device(n, dev, first_clip=0, utts_per_speaker=123)
pipe = VerificationPipeline(model.eval(), use_vad=True, normalize=True, preemph_cof=0.98, crop_rng="device", micro_batch=small)
a = pipe.embed(pcm).clone()
pipe.micro_batch = big
b = pipe.embed(pcm)
torch.cuda.synchronize()
same = bool(torch.equal(a, b))
rows = int((a != b).any(dim=1).sum())
print("micro-batches of %d vs %d clips over %d clips: %s (%d rows differ, max |diff| %.3g, finite: %s, peak memory %.1f GB)"
      % (small, big, n, "bit-identical" if same else "DIFFERENT", rows, float((a - b).abs().max()), bool(torch.isfinite(b).all()),
         torch.cuda.max_memory_allocated() / 1e9))
sys.exit(0 if same else 1)
