#!/usr/bin/env python3
"""The clock the chip holds in the bench's own kernel mix: the whole pipeline (VAD .. C3D2, 4 018-clip micro-batches) runs back to
back for ~10 s from the -DSVK_TUNING build (make -C speaker_verification_amd/csrc stamps); every few seconds ONE micro-batch runs
with the in-kernel stamps on, which print s_memtime cycles per s_memrealtime tick for the first block and conv2_2.
   python tools/clock_in_pipeline.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                  # noqa: E402
from speaker_verification_amd import _lib                                    # noqa: E402
_lib.LIB_PATH = os.environ.get("SVK_TOOL_LIB", os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "build_variants", "libsvk_stamps.so"))
from speaker_verification_amd import synth                                   # noqa: E402
from speaker_verification_amd.engine import get_engine                       # noqa: E402
from speaker_verification_amd.model import seeded_model                      # noqa: E402
from speaker_verification_amd.pipeline import VerificationPipeline           # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
eng = get_engine(0)
pcm, _ = synth.corpus_device(4018, eng.device, first_clip=0, utts_per_speaker=123)
pipe = VerificationPipeline(seeded_model(2024, n_labels=1211), use_vad=True, normalize=True, preemph_cof=0.98, crop_rng="device",
                            micro_batch=4096)
pipe.embed(pcm)
torch.cuda.synchronize()
time.sleep(2.0)
t0 = time.time()
k = 0
while time.time() - t0 < secs:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(12):
        pipe.embed(pcm)
    b.record()
    torch.cuda.synchronize()
    print("t = %5.2f s: %.2f ms per 4 018-clip micro-batch" % (time.time() - t0, a.elapsed_time(b) / 12), file=sys.stderr, flush=True)
    k += 1
    if k % 3 == 0:
        os.environ["SVK_C3D2_STAMPS"] = "1"
        pipe.embed(pcm)
        torch.cuda.synchronize()
        os.environ.pop("SVK_C3D2_STAMPS", None)
