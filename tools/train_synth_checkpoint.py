#!/usr/bin/env python3
"""Train ONE small C3D2 checkpoint on synthetic speakers, so that the benchmark's EER is an operating point and
not the coin flip of a random-init network ("today weights are random, so EER parity is a numerical,
not an accuracy, statement"; /root/reference/evaluation.py:90-146 and model.py:357 expect a trained checkpoint
that does not ship).

A tool, not product: `model.C3D2` in training mode under torch autograd on the GPU box (PyTorch-ROCm), features
from the libsvk front end (VAD -> pre-emphasis + log-mel -> CMVN, the benchmark's configuration), softmax
cross-entropy over the training speakers like /root/reference/train.py:60-99.  Training speakers are
`synth.corpus_device` speakers 2000 .. 2000 + S (the benchmark's corpus uses speakers 0 .. 1208) under another
clip seed; the held-out check embeds the benchmark's own 4 874-clip / 40-speaker verification block through the
libsvk network and prints its EER.

    python tools/train_synth_checkpoint.py --out gpurun_out/c3d2_synth.pt

writes {"state_dict", "meta"} (plain tensors / numbers / strings: `torch.load(weights_only=True)` reads it).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--speakers", type=int, default=100, help="training speakers = FC6 rows (the reference's C3D2(100, 1))")
    ap.add_argument("--utts", type=int, default=100, help="3 s clips per training speaker")
    ap.add_argument("--first-speaker", type=int, default=2000)
    ap.add_argument("--clip-seed", type=int, default=777)
    ap.add_argument("--init-seed", type=int, default=2024)
    ap.add_argument("--epochs", type=int, default=12)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--weight-decay", type=float, default=1e-4)
    ap.add_argument("--budget-s", type=float, default=420.0, help="stop after the epoch that passes this many seconds")
    ap.add_argument("--test-clips", type=int, default=4874)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "c3d2_synth.pt"))
    args = ap.parse_args()

    import torch
    import torch.nn.functional as F
    from speaker_verification_amd import constants as c, evaluation, synth
    from speaker_verification_amd.engine import get_engine
    from speaker_verification_amd.model import seeded_model
    from speaker_verification_amd.pipeline import VerificationPipeline, enroll_last_utterance

    torch.cuda.set_device(0)
    eng = get_engine(0)
    dev = eng.device
    log = open(os.path.splitext(args.out)[0] + ".log", "w")

    def say(msg):
        print(msg, flush=True)
        log.write(msg + "\n")
        log.flush()

    model = seeded_model(args.init_seed, n_labels=args.speakers).to(dev)
    pipe = VerificationPipeline(model, use_vad=True, normalize=True, preemph_cof=0.98, crop_rng="device", micro_batch=4096)

    # ---- training features (resident): VAD -> pre-emph + lmfe -> CMVN, once ----
    n_train = args.speakers * args.utts
    t0 = time.time()
    feats, frames = [], []
    for lo in range(0, n_train, 2000):
        m = min(2000, n_train - lo)
        pcm, _ = synth.corpus_device(m, dev, first_clip=args.first_speaker * args.utts + lo, utts_per_speaker=args.utts,
                                     seed=args.clip_seed)
        voiced, vlen = pipe.voiced(pcm)
        f, nf = pipe.features(voiced, vlen)
        feats.append(f)
        frames.append(nf)
        del pcm, voiced
    T = max(f.shape[1] for f in feats)
    feat = torch.zeros((n_train, T, c.NUM_COEF), device=dev)
    at = 0
    for f in feats:
        feat[at:at + f.shape[0], :f.shape[1]] = f
        at += f.shape[0]
    n_frames = torch.cat(frames)
    labels = (torch.arange(n_train, device=dev) // args.utts).long()
    ok = n_frames > c.CUBE_FRAMES
    say("features of %d clips (%d speakers) in %.1f s; %d clips too short after VAD (skipped)"
        % (n_train, args.speakers, time.time() - t0, int((~ok).sum())))
    keep = torch.nonzero(ok).flatten()
    del feats, frames

    # ---- held-out block: the benchmark's verification shape ----
    test_pcm, _ = synth.corpus_device(args.test_clips, dev, first_clip=0, utts_per_speaker=123)
    spk = (np.arange(args.test_clips) // 123).astype(np.int32)
    ids, last = enroll_last_utterance(None, spk)
    lab = (spk[:, None] == ids[None, :]).astype(np.float64)

    def held_out_eer():
        model.eval()
        pipe.refresh_model()
        emb = pipe.embed(test_pcm)
        sc = pipe.score(emb, emb[torch.from_numpy(last).to(dev)]).cpu().numpy().astype(np.float64)
        eer, auc, _, _ = evaluation.get_eer_auc(lab.flatten(), sc.flatten())
        return float(eer), float(auc)

    say("held-out EER before training (random init, uncalibrated BatchNorm): %.4f / AUC %.4f" % held_out_eer())
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    steps_per_epoch = int(keep.numel()) // args.batch
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=args.lr, total_steps=args.epochs * steps_per_epoch)
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.init_seed + 1)
    step, t_start, history = 0, time.time(), []
    for epoch in range(args.epochs):
        model.train()
        perm = keep[torch.randperm(keep.numel(), device=dev, generator=gen)]
        # fresh crop starts every epoch (utils.py:372 draws them per item): keyed by (seed + epoch, clip)
        crops = eng.draw_crops(n_frames, c.CUBE_CROPS, c.CUBE_FRAMES, 4242 + epoch, 0, pipe.bad_clips)
        run_loss, run_hit, seen = 0.0, 0, 0
        for b in range(steps_per_epoch):
            rows = perm[b * args.batch:(b + 1) * args.batch]
            cubes = eng.cube_gather(feat[rows], crops[rows], c.CUBE_FRAMES)
            emb = model(cubes, development=False)
            logits = model.FC6(model.PReLu5(emb))           # model.py:170-172 without the softmax (cross_entropy applies it)
            loss = F.cross_entropy(logits, labels[rows])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            sched.step()
            step += 1
            run_loss += float(loss.detach()) * rows.numel()
            run_hit += int((logits.argmax(1) == labels[rows]).sum())
            seen += rows.numel()
        eer, auc = held_out_eer()
        history.append({"epoch": epoch + 1, "loss": run_loss / seen, "train_acc": run_hit / seen, "eer": eer, "auc": auc,
                        "seconds": time.time() - t_start})
        say(json.dumps(history[-1]))
        if time.time() - t_start > args.budget_s:
            say("time budget reached after epoch %d" % (epoch + 1))
            break
    model.eval()
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    slopes = [float(v.min()) for k, v in state.items() if "PReLu" in k] + [float(v.max()) for k, v in state.items() if "PReLu" in k]
    meta = {"tool": "tools/train_synth_checkpoint.py", "speakers": args.speakers, "utts_per_speaker": args.utts,
            "first_speaker": args.first_speaker, "clip_seed": args.clip_seed, "init_seed": args.init_seed, "steps": step,
            "epochs": len(history), "batch": args.batch, "lr": args.lr, "weight_decay": args.weight_decay,
            "held_out_eer": history[-1]["eer"], "held_out_auc": history[-1]["auc"], "train_acc": history[-1]["train_acc"],
            "prelu_slope_range": [min(slopes), max(slopes)],
            "front_end": "energy VAD -> preemphasis(0.98) -> lmfe(16000, 0.025, 0.01, 40, 1024) of int16 / 32768 -> cmvn(variance)",
            "torch": str(torch.__version__)}
    torch.save({"state_dict": state, "meta": {k: (v if isinstance(v, (int, float, str)) else json.dumps(v)) for k, v in meta.items()}},
               args.out)
    json.dump({"meta": meta, "history": history}, open(os.path.splitext(args.out)[0] + ".json", "w"), indent=1)
    say("wrote %s (%d steps, held-out EER %.4f)" % (args.out, step, history[-1]["eer"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
