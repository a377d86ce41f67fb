#!/usr/bin/env python3
"""Stage-by-stage timing of the hot path on one GPU (HIP events), for DESIGN.md and tuning.
    python tools/profile_stages.py [--batch 1024]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    from speaker_verification_amd import synth
    from speaker_verification_amd.engine import get_engine
    from speaker_verification_amd.model import seeded_model
    from speaker_verification_amd.pipeline import VerificationPipeline
    eng = get_engine(0)
    pcm, _ = synth.corpus_device(args.batch, eng.device)
    model = seeded_model(1)
    pipe = VerificationPipeline(model, normalize=True, preemph_cof=0.98, crop_rng="device",
                                micro_batch=args.batch)

    def run():
        marks = [("start", torch.cuda.Event(enable_timing=True))]
        marks[0][1].record()

        def mark(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))
        voiced, vlen = pipe.voiced(pcm)
        mark("vad+compact")
        feat, nf, _ = eng.features(voiced, pipe.spec, lengths=vlen)
        mark("frontend(lmfe B)")
        eng.cmvn_(feat, nf, variance=True)
        mark("cmvn")
        idx = eng.draw_crops(nf, 20, 80, 1, 0, pipe.bad_clips)
        cube = pipe.cubes(feat, idx)
        mark("crops+cube")
        emb = pipe.embed_cubes(cube)
        mark("C3D2 (torch)")
        pipe.score(emb[:min(4874, len(emb))], emb[:40])
        mark("cosine")
        torch.cuda.synchronize()
        return {marks[i][0]: marks[i - 1][1].elapsed_time(marks[i][1]) for i in range(1, len(marks))}

    run()
    rows = [run() for _ in range(args.reps)]
    med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    total = sum(med.values())
    out = {"batch": args.batch, "ms": med, "total_ms": total, "utt_per_s": args.batch / total * 1e3,
           "mean_voiced_frames": float(eng.features(pipe.voiced(pcm)[0], pipe.spec,
                                                    lengths=pipe.voiced(pcm)[1])[1].float().mean().item())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
