#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X (contract: see the task prompt / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W

One STEP = one pass of the whole path over this rank's shard of synthetic clips:
    int16 PCM (resident in HBM) -> energy VAD -> fused pre-emphasis + log-mel(40) front end -> CMVN
    -> 20x80x40 cube -> C3D2 embedding (PyTorch-ROCm, f32) -> all-gather of the [clips,128] shards
    (RCCL) -> 4 874 x 40 cosine score matrix (MFMA).
Per-rank work is fixed (weak scaling): 18 581 clips = ceil(148 642 / 8), the per-GPU shard of
BASELINE.json's 148 642-clip corpus.  value = clips processed by all ranks / max-over-ranks time.

The JSON line also carries
  roofline      -- the fused front-end kernel (the dominant hand-written kernel): algorithmic
                   HBM bytes / launch over its HIP-event duration inside the timed region;
  frontend_A    -- BASELINE config 2 measured beside it: 1 024 clips, SpeechPy defaults
                   (pre-emph + MFCC-13 + CMVN), utterances/s and roofline fraction;
  cpu_baseline  -- the CPU oracle (NumPy/torch-CPU restatement of the reference, kind "port")
                   timed on a bounded sample on rank 0, N = 1 only;
  parity        -- GPU vs oracle on that sample + EER of the 4 874 x 40 score matrix.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)
N_TEST, N_TEST_SPK = 4874, 40    # VoxCeleb1 verification split (README.md:4-7)
UTTS_PER_SPK = 123


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=18581, help="clips per rank and step")
    ap.add_argument("--micro-batch", type=int, default=1024)
    ap.add_argument("--cpu-sample", type=int, default=48, help="clips of the CPU-oracle baseline (0 = skip)")
    ap.add_argument("--no-vad", action="store_true")
    ap.add_argument("--no-cmvn", action="store_true")
    ap.add_argument("--no-preemph", action="store_true")
    ap.add_argument("--no-channels-last", action="store_true")
    ap.add_argument("--frontend-only", action="store_true", help="time BASELINE config 2 only (for rocprof)")
    return ap.parse_args()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/rNN_frontend_pmc.json, written by tools/summarize_prof.py from separate --pmc
    passes over `bench.py --frontend-only`: 1 024 full 3 s clips per launch, FETCH_SIZE doubled
    per the gfx950 correction).  None when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_frontend_pmc.json")))
    if not files:
        return None, None
    try:
        rec = json.load(open(files[-1])).get(kernel, {})
    except (OSError, ValueError):
        return None, None
    return rec.get("hbm_traffic_bytes_per_launch"), os.path.basename(files[-1])


def frontend_A_bench(eng, torch, reps=20, n_clips=1024):
    """BASELINE config 2: batched MFCC pipeline, 1 024 x 3 s clips, SpeechPy defaults."""
    from speaker_verification_amd import _lib, synth
    from speaker_verification_amd.engine import spec_from_seconds
    spec = spec_from_seconds(16000, 0.020, 0.01, 512, 40, 13, _lib.OUT_MFCC, preemph=True, preemph_cof=0.98)
    base = np.stack([synth.noise_clip(s) for s in range(16)])
    pcm = eng.to_device(np.tile(base, (n_clips // 16, 1)))
    for _ in range(3):
        feat, nf, _ = eng.features(pcm, spec)
        eng.cmvn_(feat, nf, variance=True)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
    for a, b, c in ev:
        a.record()
        feat, nf, _ = eng.features(pcm, spec)
        b.record()
        eng.cmvn_(feat, nf, variance=True)
        c.record()
    torch.cuda.synchronize()
    t_fe = float(np.median([a.elapsed_time(b) for a, b, _ in ev])) * 1e-3
    t_all = float(np.median([a.elapsed_time(c) for a, _, c in ev])) * 1e-3
    bytes_per_utt = 48000 * 2 + 298 * 13 * 4          # 111 496 B
    gbs = n_clips * bytes_per_utt / t_fe / 1e9
    traffic, src = pmc_traffic("frontend_kernel<int16,nfft512>")
    return {"workload": "configs[1]: 1024 x 3 s clips, pre-emph + MFCC-13 (nfft 512) + CMVN",
            "utt_per_s": n_clips / t_all, "frontend_kernel_ms": t_fe * 1e3, "with_cmvn_ms": t_all * 1e3,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                         "algorithmic_bytes_per_launch": n_clips * bytes_per_utt,
                         "kernel": "frontend_kernel<int16,nfft512>", "bytes_per_utt": bytes_per_utt}}


def frontend_B_bench(eng, torch, reps=20, n_clips=1024):
    """The model's front end on its own: 1 024 x 3 s clips, pre-emph + lmfe(25 ms / 1024 / 40)."""
    from speaker_verification_amd import _lib, synth
    from speaker_verification_amd.engine import spec_from_seconds
    spec = spec_from_seconds(16000, 0.025, 0.01, 1024, 40, 40, _lib.OUT_LMFE, preemph=True, preemph_cof=0.98)
    base = np.stack([synth.noise_clip(s) for s in range(16)])
    pcm = eng.to_device(np.tile(base, (n_clips // 16, 1)))
    for _ in range(3):
        eng.features(pcm, spec)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(reps)]
    for a, b in ev:
        a.record()
        eng.features(pcm, spec)
        b.record()
    torch.cuda.synchronize()
    t = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e-3
    bytes_per_utt = 48000 * 2 + 297 * 40 * 4
    gbs = n_clips * bytes_per_utt / t / 1e9
    traffic, src = pmc_traffic("frontend_kernel<int16,nfft1024>")
    return {"workload": "1024 x 3 s clips, pre-emph + lmfe-40 (nfft 1024)", "utt_per_s": n_clips / t,
            "frontend_kernel_ms": t * 1e3,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                         "algorithmic_bytes_per_launch": n_clips * bytes_per_utt,
                         "kernel": "frontend_kernel<int16,nfft1024>", "bytes_per_utt": bytes_per_utt}}


def stage_breakdown(pipe, eng, torch, chunk, first_utt):
    """HIP-event time of every stage of ONE micro-batch (serialised, outside the timed region) and
    the algorithmic HBM rate of each hand-written kernel (bytes as in DESIGN.md section 3)."""
    from speaker_verification_amd import constants as c
    n, L = chunk.shape

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts)), out

    rows = {}
    t, (voiced, vlen) = timed(lambda: pipe.voiced(chunk))
    kept = float(vlen.float().sum().item()) if vlen is not None else float(n * L)
    rows["vad+compact"] = (t, n * L * 2 + kept * 2)
    t, (feat, nf, _) = timed(lambda: eng.features(voiced, pipe.spec, lengths=vlen))
    frames = float(nf.float().sum().item())
    rows["frontend (lmfe-40, nfft 1024)"] = (t, kept * 2 + frames * 40 * 4)
    t, _ = timed(lambda: eng.cmvn_(feat, nf, variance=True))
    rows["cmvn"] = (t, 2 * frames * 40 * 4)
    t, idx = timed(lambda: eng.draw_crops(nf, c.CUBE_CROPS, c.CUBE_FRAMES, pipe.crop_seed, first_utt, pipe.bad_clips))
    rows["draw_crops"] = (t, n * (4 + 80))
    geo = pipe.embedder.first_layer_windows(c.CUBE_CROPS, 40) if pipe.embedder is not None else None
    if geo is not None:
        kd, kw, G = geo
        t, win = timed(lambda: eng.cube_windows(feat, idx, c.CUBE_FRAMES, kd, kw, G))
        rows["cube as first-layer patch matrix"] = (t, frames * 40 * 4 + float(win.numel()) * 4)
        t, emb = timed(lambda: pipe.embedder.from_windows(win, n, c.CUBE_CROPS, c.CUBE_FRAMES, 40))
    else:
        t, cube = timed(lambda: pipe.cubes(feat, idx))
        rows["cube_gather"] = (t, frames * 40 * 4 + n * 256000)
        t, emb = timed(lambda: pipe.embed_cubes(cube))
    rows["C3D2 forward (PyTorch-ROCm)"] = (t, None)
    t, _ = timed(lambda: pipe.score(emb, emb[:40]))
    rows["cosine %dx40" % n] = (t, (n + 40) * 128 * 4 + n * 40 * 4)
    total = sum(v[0] for v in rows.values())
    return {"clips": n, "total_ms": total,
            "stages": {k: {"ms": v[0], "share": v[0] / total,
                           "algorithmic_GBps": None if v[1] is None else v[1] / (v[0] * 1e-3) / 1e9}
                       for k, v in rows.items()}}


def cosine_mfma_bench(eng, torch, reps=10):
    """The all-pairs cosine kernel at dev-set scale (148 642 x 1 211 x 128, SURVEY 8d's stress shape):
    its MFMA utilisation against the dense f32 matrix peak (157.3 TFLOP/s).  The verification shape
    (4 874 x 40) is 50 MFLOP -- launch-latency-bound by construction -- and is timed inside every step."""
    nt, ns, d = 148642, 1211, 128
    t = torch.randn(nt, d, device=eng.device)
    e = torch.randn(ns, d, device=eng.device)
    for _ in range(3):
        eng.cosine_scores(t, e)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.cosine_scores(t, e)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    tf = 2.0 * nt * ns * d / ms / 1e9
    return {"workload": "%d x %d x %d cosine score matrix" % (nt, ns, d), "ms": ms,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
                         "traffic": None, "kernel": "cosine_tiled_kernel<true> (v_mfma_f32_16x16x4_f32)",
                         "output_GBps": nt * ns * 4 / ms / 1e6}}


def ingest_bench(eng, torch, reps=10, n_clips=1024):
    """svk_ingest_resample: 1 024 x 3 s mono clips at 48 kHz -> 16 kHz float32.
    Algorithmic bytes per clip: 144 000 x 2 read + 48 000 x 4 written = 480 000 B."""
    from speaker_verification_amd import ingest
    pcm = (torch.randn(n_clips, 144000, device=eng.device) * 3000).to(torch.int16)
    up, down = ingest.rational_ratio(48000, 16000)
    taps = ingest.resample_taps(up, down)
    for _ in range(3):
        eng.resample(pcm, up, down, taps)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.resample(pcm, up, down, taps)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    gbs = n_clips * 480000 / ms / 1e6
    return {"workload": "%d x 3 s clips, 48 kHz mono int16 -> 16 kHz float32" % n_clips, "ms": ms,
            "utt_per_s": n_clips / ms * 1e3,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": None, "kernel": "decimate_kernel<3, float>",
                         "bytes_per_utt": 480000}}


def cpu_baseline(pcm_host, crop_idx, state, preemph, cmvn, use_vad, passes=3):
    """The oracle (kind 'port') doing exactly the reference's per-utterance sequence on the host:
    vad -> preemphasis -> lmfe -> cmvn -> cube -> C3D2 at batch 1 -> per-pair cosine.  The sample is
    walked `passes` times (about 10 s of host work); the embeddings of the last pass are returned."""
    import torch
    from oracle import model_ref, scoring_ref, speechpy_ref, vad_ref
    from speaker_verification_amd import constants as c
    t0 = time.perf_counter()
    for _ in range(max(1, passes)):
        embs = _cpu_pass(pcm_host, crop_idx, state, preemph, cmvn, use_vad, model_ref, scoring_ref, speechpy_ref,
                         vad_ref, c)
    dt = (time.perf_counter() - t0) / max(1, passes)
    return embs, dt, torch.get_num_threads()


def _cpu_pass(pcm_host, crop_idx, state, preemph, cmvn, use_vad, model_ref, scoring_ref, speechpy_ref, vad_ref, c):
    embs = []
    for i in range(pcm_host.shape[0]):
        clip = pcm_host[i]
        if use_vad:
            _, _, clip = vad_ref.vad_energy(clip, c.SAMPLE_RATE, c.VAD_FRAME_MS, c.VAD_PADDING_MS,
                                            c.VAD_ENERGY_THRESHOLD)
        sig = speechpy_ref.preemphasis(clip, cof=0.98) if preemph else clip
        feat = speechpy_ref.lmfe(sig, c.SAMPLE_RATE, c.FRAME_LEN, c.FRAME_STEP, c.NUM_COEF, c.NUM_FFT)
        if cmvn:
            feat = speechpy_ref.cmvn(feat, variance_normalization=True)
        cube = model_ref.feature_cube(feat, crop_idx[i])[None]
        embs.append(model_ref.c3d2_embed(state, cube).numpy()[0])
    embs = np.stack(embs)
    enroll = embs[::max(1, len(embs) // 8)]
    for i in range(len(embs)):
        scoring_ref.compute_similarity(embs[i], enroll)
    return embs


def main():
    args = parse()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # one MIOpen find-db / kernel cache per rank: N processes searching kernels at once would
        # otherwise contend for the same sqlite files under ~/.config/miopen and ~/.cache/miopen
        import tempfile
        tag = os.path.join(tempfile.gettempdir(), "svk_miopen_rank%s" % os.environ.get("LOCAL_RANK", "0"))
        os.makedirs(tag, exist_ok=True)
        os.environ.setdefault("MIOPEN_USER_DB_PATH", tag)
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", tag)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("SVK_BENCH_FORCE_DIST") == "1"   # the latter: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints a version banner on STDOUT when the communicator comes up; stdout is reserved
        # for the one JSON line, so fd 1 points at stderr until the first collective has run.
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from speaker_verification_amd import constants as c, distributed as svdist, evaluation, synth
    from speaker_verification_amd.engine import get_engine
    from speaker_verification_amd.model import calibrate_batchnorm, seeded_model
    from speaker_verification_amd.pipeline import VerificationPipeline, enroll_last_utterance

    eng = get_engine(local_rank)
    dev = eng.device

    if args.frontend_only:
        res = {"A": frontend_A_bench(eng, torch, reps=max(args.steps, 5)),
               "B": frontend_B_bench(eng, torch, reps=max(args.steps, 5)),
               "ingest_resample": ingest_bench(eng, torch)}
        print(json.dumps(res))
        return

    n_local = args.clips
    n_total = n_local * world
    pcm, speakers_local = synth.corpus_device(n_local, dev, first_clip=rank * n_local, utts_per_speaker=UTTS_PER_SPK)
    model = seeded_model(2024, n_labels=1211)
    pipe = VerificationPipeline(model, use_vad=not args.no_vad, normalize=not args.no_cmvn,
                                preemph_cof=None if args.no_preemph else 0.98, crop_rng="device",
                                micro_batch=args.micro_batch, channels_last=not args.no_channels_last,
                                overlap_front=os.environ.get("SVK_BENCH_OVERLAP", "0") == "1")
    # random-init weights (no checkpoint ships) with BatchNorm statistics calibrated on 256 clips of
    # rank 0's shard, identically on every rank (model.calibrate_batchnorm explains why)
    cal_pcm, _ = synth.corpus_device(256, dev, first_clip=0, utts_per_speaker=UTTS_PER_SPK)
    _, cal = pipe.embed(cal_pcm, return_intermediates=True)
    calibrate_batchnorm(pipe.model, torch.cat([d["cube"] for d in cal]))
    pipe.refresh_model()
    del cal_pcm, cal
    n_test = min(N_TEST, n_total)
    spk_all = (np.arange(n_total) // UTTS_PER_SPK).astype(np.int32)
    ids, last = enroll_last_utterance(None, spk_all[:n_test])                  # Q17: last utterance enrols
    last_dev = torch.from_numpy(last).to(dev)

    fe_events = []

    def one_step(record):
        # timed region: everything from resident PCM to the score matrix
        local = torch.empty((n_local, 128), dtype=torch.float32, device=dev)
        for lo, hi in pipe.chunks(n_local):
            chunk = pcm[lo:hi]
            voiced, vlen = pipe.voiced(chunk)
            if record:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            feat, n_frames, _ = eng.features(voiced, pipe.spec, lengths=vlen)
            if record:
                b.record()
                fe_events.append((a, b, hi - lo))
            if pipe.normalize:
                eng.cmvn_(feat, n_frames, variance=True)
            idx = eng.draw_crops(n_frames, c.CUBE_CROPS, c.CUBE_FRAMES, pipe.crop_seed, rank * n_local + lo,
                                 pipe.bad_clips)
            local[lo:hi] = pipe.embed_features(feat, idx)      # cube (as the first layer's patch matrix) -> C3D2
        full = svdist.all_gather_embeddings(local, n_total)
        scores = pipe.score(full[:n_test], full[:n_test][last_dev])
        return full, scores

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full, scores = one_step(True)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    bad = int(pipe.bad_clips.item())

    result = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_total * args.steps / dt
        # roofline of the fused front-end kernel, from HIP events on the launch stream
        fe_ms = np.array([a.elapsed_time(b) for a, b, _ in fe_events])
        fe_clips = np.array([n for _, _, n in fe_events])
        bytes_per_utt = 48000 * 2 + 297 * 40 * 4                 # SURVEY 8(d) front end B: 143 520 B
        avg_launch_s = float(fe_ms.mean()) * 1e-3
        gbs = float(fe_clips.mean()) * bytes_per_utt / avg_launch_s / 1e9
        traffic, traffic_src = pmc_traffic("frontend_kernel<int16,nfft1024>")
        labels = (spk_all[:n_test, None] == ids[None, :]).astype(np.float64)
        sc = scores.cpu().numpy().astype(np.float64)
        eer, auc, _, _ = evaluation.get_eer_auc(labels.flatten(), sc.flatten())
        eer_dev, auc_dev = evaluation.get_eer_auc_device(labels, scores)       # svk_roc_eer on the same matrix
        result = {
            "metric": "utterances/sec (MFCC->embed->cosine)", "value": value, "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (seeded formant 'voices', 3 s / 16 kHz int16, generated on device; random-init C3D2)",
            "config": {"workload": "configs[4] per-GPU shard: %d clips/rank x %d rank(s): energy-VAD -> pre-emph + "
                                   "lmfe(25ms/10ms/1024/40) -> CMVN -> 20x80x40 cube -> C3D2(f32) -> all-gather -> "
                                   "%dx%d cosine" % (n_local, world, n_test, len(ids)),
                       "clips_per_rank": n_local, "micro_batch": pipe.chunks(n_local)[0][1], "vad": pipe.use_vad,
                       "cmvn": pipe.normalize, "preemph": not args.no_preemph, "parallelism": "dp%d" % world,
                       "crop_rng": "device"},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_note": "PMC pass ran this kernel on 1 024 FULL 3 s clips per launch "
                                         "(algorithmic 146 964 480 B); launches here carry VAD-shortened clips",
                         "kernel": "frontend_kernel<int16,nfft1024>", "avg_launch_ms": avg_launch_s * 1e3,
                         "clips_per_launch": float(fe_clips.mean()), "bytes_per_utt": bytes_per_utt,
                         "frontend_share_of_step": float(fe_ms.sum()) / args.steps / ms_per_step},
            "eer": {"eer": eer, "auc": auc, "eer_device": eer_dev, "auc_device": auc_dev, "pairs": int(labels.size),
                    "short_clips": bad},
        }

    if rank == 0 and world == 1:
        # host-fed variant (PCIe included; NOT `value`): 4 micro-batches of the shard from pageable host memory
        n_host = min(n_local, 4 * pipe.chunks(n_local)[0][1])
        host_pcm = pcm[:n_host].cpu().numpy()
        pipe.embed_host(host_pcm)
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        emb_h = pipe.embed_host(host_pcm)
        torch.cuda.synchronize()
        t_h = time.perf_counter() - t_h
        pinned_pcm = torch.from_numpy(host_pcm).pin_memory()
        pipe.embed_host(pinned_pcm)
        torch.cuda.synchronize()
        t_p = time.perf_counter()
        emb_p = pipe.embed_host(pinned_pcm)
        torch.cuda.synchronize()
        t_p = time.perf_counter() - t_p
        result["host_fed"] = {"utt_per_s": n_host / t_h, "utt_per_s_from_pinned": n_host / t_p, "clips": n_host,
                              "max_abs_diff_vs_resident": float(max((emb_h - full[:n_host]).abs().max().item(),
                                                                    (emb_p - full[:n_host]).abs().max().item())),
                              "note": "pageable int16 NumPy -> 8-thread staging into a pinned double buffer -> copy "
                                      "stream -> same kernels; from_pinned: the caller's buffer is already pinned"}
        del pinned_pcm
        lo0, hi0 = pipe.chunks(n_local)[0]
        result["micro_batch_breakdown"] = stage_breakdown(pipe, eng, torch, pcm[lo0:hi0], 0)
        result["frontend_A"] = frontend_A_bench(eng, torch)
        result["cosine_mfma"] = cosine_mfma_bench(eng, torch)
        result["ingest_resample"] = ingest_bench(eng, torch)
        if args.cpu_sample > 0:
            ns = min(args.cpu_sample, n_local)
            # three utterances each of ns/3 speakers spread over the shard (the last one enrols, Q17)
            n_spk_s = max(1, ns // 3)
            spk_step = max(1, (n_local // UTTS_PER_SPK) // n_spk_s)
            pick = np.array([min(n_local - 1, (k * spk_step) * UTTS_PER_SPK + u) for k in range(n_spk_s)
                             for u in range(3)], dtype=np.int64)
            ns = pick.size
            sample = pcm[torch.from_numpy(pick).to(dev)]
            emb, inter = pipe.embed(sample, return_intermediates=True)
            crops = np.zeros((ns, c.CUBE_CROPS), dtype=np.int32)
            for d in inter:
                crops[d["lo"]:d["hi"]] = d["crop_idx"].cpu().numpy()
            state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            ref_emb, cpu_dt, threads = cpu_baseline(sample.cpu().numpy(), crops, state, not args.no_preemph,
                                                    pipe.normalize, pipe.use_vad)
            got = emb.cpu().numpy()
            from oracle import scoring_ref
            spk_s = spk_all[pick]
            lab = (spk_s[:, None] == np.unique(spk_s)[None, :]).astype(np.float64)
            _, last_s = enroll_last_utterance(None, spk_s)
            s_gpu = pipe.score(emb, emb[torch.from_numpy(last_s).to(dev)]).cpu().numpy().astype(np.float64)
            s_ref = scoring_ref.cosine_matrix(ref_emb, ref_emb[last_s]).astype(np.float64)
            par = {"sample_clips": ns, "embed_max_abs_diff": float(np.abs(got - ref_emb).max()),
                   "embed_scale": float(np.abs(ref_emb).max()),
                   "score_max_abs_diff": float(np.abs(s_gpu - s_ref).max())}
            if lab.shape[1] > 1:
                par["eer_gpu"] = float(evaluation.get_eer_auc(lab.flatten(), s_gpu.flatten())[0])
                par["eer_cpu_ref"] = float(scoring_ref.get_eer_auc(lab.flatten(), s_ref.flatten())[0])
            # full-size check: oracle cosine + EER on the SAME embeddings must give the same EER
            fe = full[:n_test].cpu().numpy()
            s_or = scoring_ref.cosine_matrix(fe, fe[last]).astype(np.float64)
            par["full_matrix_score_max_abs_diff"] = float(np.abs(s_or - sc).max())
            par["full_matrix_eer_cpu_ref"] = float(scoring_ref.get_eer_auc(labels.flatten(), s_or.flatten())[0])
            result["parity"] = par
            result["cpu_baseline"] = {"value": ns / cpu_dt, "unit": "utterances/s", "cores": threads, "kind": "port",
                                      "sample": "%d clips spread over the same shard through oracle/ (vad -> preemph -> lmfe "
                                                "-> cmvn -> cube -> C3D2 batch 1 -> per-pair cosine), 3 passes of %.1f s" %
                                                (ns, cpu_dt)}
    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
