#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X (contract: see the task prompt / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W

One STEP = one pass of the whole path over BASELINE.json's corpus, 148 642 synthetic 3 s / 16 kHz clips,
sharded over the N ranks (rank r owns the contiguous range of ceil(148 642 / N) clips; STRONG scaling:
the total is fixed), PCM resident in HBM:
    int16 PCM -> energy VAD (an index of the voiced frames) -> fused pre-emphasis + log-mel(40) front end reading through it
    -> CMVN -> 20x80x40 cube (never materialised) -> C3D2 embedding (seven libsvk MFMA kernels: svk_c3d2_stage1 / stage2 /
    conv31 / conv32t / conv41 / conv42 / fc5) -> all-gather of the [clips,128] shards (RCCL) -> 4 874 x 40 cosine score matrix (MFMA).
value = 148 642 clips x steps / max-over-ranks time.  Nothing in the step is a framework operator: torch supplies
device memory, streams, events and torch.distributed.  Weights: the committed checkpoint trained on synthetic speakers that
are not in the corpus (speaker_verification_amd/checkpoints/c3d2_synth.pt; EER 1.7 % on the verification block), or with
--random-init the seeded random-init network of rounds 1 - 3 (EER ~ 0.5).

Launch: with WORLD_SIZE in the environment (torch.distributed.run) this process is one rank; without it
and --gpus N > 1 this process only starts N fresh rank processes of itself (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set), polls them, ends the others when one dies, relays rank 0's JSON line.

The JSON line also carries
  roofline      -- the dominant kernel, c3d2_stage1h_kernel (cube + conv1_1 + conv1_2 + pool1 through two-piece f16 products,
                   ~30 % of the step), bound MFMA: frac = ISSUED matrix work (SQ_INSTS_MFMA per cube from the committed PMC
                   pass x 16 384 FLOP x cubes per launch) / the HIP-event duration of each launch in the timed region / the
                   dense f16 matrix peak (16 x 157.3 TFLOP/s) -- a share of the F16 matrix pipe's issue slots, never above 1;
                   algorithmic_frac = SURVEY 8(d)'s direct-form f32 multiply-adds against the F32 matrix peak instead;
                   counters_from / stale: what the committed counters were collected from (sha256 of the kernel sources,
                   of libsvk.so, git HEAD) and whether that differs from the sources this run uses;
  roofline_network -- the same two fractions for EVERY network kernel (stage2 = conv2_1 + conv2_2, conv3_1, conv3_2,
                   conv4_1, conv4_2, fc5) with each one's share of the step; roofline_stage2 = its stage2 row;
                   each row names its pipe and peak (conv1_1 .. conv4_1: f16, conv4_2 and fc5: f32);
                   valu_per_mfma / fp32_lanes_busy: the other vector instructions per MFMA (committed SQ_INSTS_VALU)
                   and, for the f32 rows, frac x (1 + valu_per_mfma x 4 / 32) -- f32 MFMA and f32 VALU never co-execute on
                   this chip, so this is the share of the SIMDs' FP32 issue slots that is occupied at all (the rest are stalls);
  roofline_frontend -- the fused front-end kernel (HBM roof): algorithmic HBM bytes of the launches it actually
                   ran (VAD-shortened clips) over their HIP-event durations;
  roofline_e2e  -- the whole step: frac = the time the matrix pipes need at their peaks for the MFMA work issued per
                   utterance / wall time; algorithmic_frac from SURVEY 8(d)'s 676.6 MFLOP against the f32 matrix peak
                   (ceiling 232 k utt/s/GPU on that pipe);
  ranks_seen / backend / allgather_us / per_rank_ms / slowest_rank / fastest_rank / scaling_efficiency_vs -- what
                   torch.distributed reports, the HIP-event time of the embedding all-gather, every rank's own step time,
                   N x the committed 1-GPU value for reference;
  frontend_A, cosine_mfma, stage_kernels, ingest_resample, ragged -- the other hand-written kernels / workloads on their own;
  cpu_baseline  -- the CPU oracle (NumPy/torch-CPU restatement of the reference, kind "port") timed in a
                   fresh child process on rank 0, N = 1 only: the per-utterance chain on single-threaded workers at
                   pool sizes around the container's CPU quota (`value` = the best; every worker warmed before the clock
                   starts; chain and pair-by-pair scoring timed separately), plus 1-core / all-core and batch-1 / batch-64
                   variants per stage;
  parity        -- the PRODUCTION path (libsvk end to end) vs the oracle on that sample: embeddings, scores, the EER on
                   both sides (eer_equal) + the EER of the 4 874 x 40 matrix through host sklearn, svk_roc_eer and the oracle.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)
F32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA peak
C3D2_GFLOP_PER_UTT = 0.6766      # 338.3 M multiply-adds per cube
STAGE1_GFLOP_PER_UTT = 2 * (12.4416 + 143.327232) / 1e3   # conv1_1 12.44 M + conv1_2 143.33 M multiply-adds
STAGE2_GFLOP_PER_UTT = 2 * (46.44864 + 66.3552) / 1e3     # conv2_1 46.45 M + conv2_2 66.36 M
MFMA_FLOP = 2048                 # v_mfma_f32_16x16x4_f32: 16 x 16 x 4 multiply-adds
F16_MATRIX_PEAK_TFLOPS = 16 * F32_MATRIX_PEAK_TFLOPS   # "~2.5 PF dense": v_mfma_f32_16x16x32_f16 issues every 16 cycles with 8 x the K
F16_MFMA_FLOP = 16384            # v_mfma_f32_16x16x32_f16: 16 x 16 x 32 multiply-adds
# The network kernels as the pipeline times them (HIP-event spans): span name, kernel symbols in the rocprofv3 / PMC
# summaries, SURVEY 8(a)'s direct-form multiply-adds per cube (millions), and the MFMA wave-instructions the kernel issues
# per cube BY CONSTRUCTION (items x tiles x taps x 4-deep steps) -- replaced at run time by the committed
# SQ_INSTS_MFMA counter of profiles/rNN_frontend_pmc.json when that file has the kernel (they agree to 1e-4).
# conv1_1 .. conv4_1 run on the F16 matrix pipe through two-piece products (three f16 products per f32 product): their rows carry
# that pipe's FLOP per MFMA and peak; the others the f32 pipe's.
NETWORK_KERNELS = (
    ("stage1", ("c3d2_stage1h_kernel",), 12.4416 + 143.327232, 36 * (100 * 2 + 36 * 41), F16_MFMA_FLOP, F16_MATRIX_PEAK_TFLOPS),   # 36 items x (100 conv1_1 tiles x 2 + 36 conv1_2 tiles x (13 tap pairs x 3 + the last tap x 2))
    ("stage2", ("c3d2_conv21h_kernel", "c3d2_conv22h_kernel"), 46.44864 + 66.3552, 9 * 49 * 36 + 21 * 8 * 2 * 72, F16_MFMA_FLOP, F16_MATRIX_PEAK_TFLOPS),   # conv2_1: 9 items x 49 tiles (14 of 15 columns: pool2 kills the last) x 6 pairs x 3 x 2 N tiles; conv2_2: 21 items x 8 tiles x 2 N tiles x 24 taps x 3
    ("conv3_1", ("c3d2_conv31h_kernel",), 13.824, 5 * 10 * 4 * 27, F16_MFMA_FLOP, F16_MATRIX_PEAK_TFLOPS),   # 5 items x 10 tiles x 4 N tiles x 9 taps x 3
    ("conv3_2", ("c3d2_conv32h_kernel",), 30.96576, 5 * 5 * 4 * 126, F16_MFMA_FLOP, F16_MATRIX_PEAK_TFLOPS),   # 5 items x 5 tiles x 4 N tiles x 21 taps x 2 K blocks x 3
    ("conv4_1", ("c3d2_conv41h_kernel",), 11.943936, 11 * 8 * 54, F16_MFMA_FLOP, F16_MATRIX_PEAK_TFLOPS),   # 11 tiles x 8 N tiles x 9 taps x 2 K blocks x 3
    ("conv4_2", ("c3d2_tail_kernel<Conv42>",), 12.386304, 8064, MFMA_FLOP, F32_MATRIX_PEAK_TFLOPS),
    ("fc5", ("fc5_kernel",), 0.589824, 576, MFMA_FLOP, F32_MATRIX_PEAK_TFLOPS),                        # 4 K ranges x 72 steps x 16 x 8 waves / 64 cubes
)
N_CORPUS = 148642                # VoxCeleb1 dev utterances (README.md:5-7) = BASELINE configs[4]
N_TEST, N_TEST_SPK = 4874, 40    # VoxCeleb1 verification split (README.md:4-7)
UTTS_PER_SPK = 123


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--corpus", type=int, default=N_CORPUS, help="clips in the whole job (all ranks), per step")
    ap.add_argument("--micro-batch", type=int, default=0,
                    help="clips per launch sequence; 0 (default) = the shard in as few equal sequences as fit in 0.6 of the free HBM at ~2 MB of live "
                         "intermediates per clip (one MI355X: the 148 642-clip corpus in two sequences of 74 321, 146 GB at the "
                         "peak).  Measured on one box: 4 096 -> 402 k utt/s, 16 384 -> 412 k, 32 768 -> 419 k; another: 32 768 -> "
                         "415.6 k, 74 321 -> 421.8 k -- every launch of a persistent kernel drains and refills the chip; the "
                         "embeddings do not depend on the batching, bit for bit (tools/check_micro_batch.py)")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="clips of the CPU-oracle baseline (0 = skip)")
    ap.add_argument("--no-vad", action="store_true")
    ap.add_argument("--no-cmvn", action="store_true")
    ap.add_argument("--no-preemph", action="store_true")
    ap.add_argument("--random-init", action="store_true",
                    help="seeded random-init C3D2 with calibrated BatchNorm (rounds 1-3) instead of the committed trained checkpoint")
    ap.add_argument("--checkpoint", default=os.path.join(REPO, "speaker_verification_amd", "checkpoints", "c3d2_synth.pt"),
                    help="{'state_dict': ...} of a C3D2 (reference format, model.py:177-186); loaded weights-only")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-kernel side benches (profiling runs)")
    ap.add_argument("--parity-only", action="store_true",
                    help="of the extras, only the CPU-oracle leg: cpu_baseline + parity (EER on both sides) on --cpu-sample clips")
    ap.add_argument("--frontend-only", action="store_true", help="time BASELINE config 2 only (for rocprof)")
    ap.add_argument("--stages-only", action="store_true", help="time the stage-level kernels only (for rocprof)")
    ap.add_argument("--ragged-only", action="store_true", help="the realistic-length workload only (4 .. 145 s clips)")
    ap.add_argument("--c3d2-only", action="store_true",
                    help="features of 1 024 clips once, then the network (seven libsvk kernels) K times (for rocprof --pmc)")
    ap.add_argument("--backend", default=os.environ.get("SVK_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="gloo: rehearse N ranks on ONE GPU (RCCL refuses two ranks on a device)")
    ap.add_argument("--selftest", action="store_true",
                    help="launcher / collective logic only, on CPU with gloo (tests/test_distributed_cpu.py)")
    ap.add_argument("--cpu-child", default=None, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N  ->  N fresh rank processes (never a re-exec of this one)
# --------------------------------------------------------------------------------------------------
RANK_LOG_DIR = os.path.join(REPO, "gpurun_out", "bench_ranks")   # gpurun merges gpurun_out/ back: the logs survive the box
INIT_TIMEOUT_S = 120                                               # torch's default is 10 min (RCCL) / 30 min (gloo)


def _tail(path, lines=30):
    try:
        with open(path, "rb") as fh:
            return b"\n".join(fh.read().splitlines()[-lines:]).decode("utf-8", "replace")
    except OSError:
        return ""


def launch_ranks(n, argv, poll_s=0.2, grace_s=5.0):
    """N fresh rank processes of this script.  Every rank's stderr (and rank 0's stdout) goes to a file under
    gpurun_out/bench_ranks/ (a temporary directory when that is not writable); all children are polled, and the FIRST
    one that exits non-zero ends the run: the others are terminated (they would otherwise sit in init_process_group or a
    barrier until torch's timeout), that rank's last stderr lines are printed, and the launcher exits non-zero within
    seconds.  Rank 0's one JSON line is relayed only when every rank succeeded."""
    import signal
    import socket
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    log_dir = RANK_LOG_DIR
    try:
        os.makedirs(log_dir, exist_ok=True)
        open(os.path.join(log_dir, ".w"), "w").close()
        os.unlink(os.path.join(log_dir, ".w"))
    except OSError:
        log_dir = tempfile.mkdtemp(prefix="svk_bench_ranks_")
    procs, files = [], []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        err = open(os.path.join(log_dir, "rank%d.stderr" % rank), "wb")
        out = open(os.path.join(log_dir, "rank%d.stdout" % rank), "wb")
        files += [err, out]
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out, stderr=err))
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None:
            break
        time.sleep(poll_s)
    if failed is not None:
        # a dead rank leaves its peers waiting in a collective: end them now (children we started, by PID)
        for r, p in enumerate(procs):
            if codes[r] is None:
                p.send_signal(signal.SIGTERM)
        deadline = time.time() + grace_s
        for r, p in enumerate(procs):
            if codes[r] is None:
                try:
                    codes[r] = p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    codes[r] = p.wait()
    for f in files:
        f.close()
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with code %d; the other ranks were terminated (codes %s).  Last lines of "
                         "%s/rank%d.stderr:\n%s\n" % (failed, codes[failed], codes, log_dir, failed,
                                                     _tail(os.path.join(log_dir, "rank%d.stderr" % failed))))
        return 1
    sys.stderr.write(_tail(os.path.join(log_dir, "rank0.stderr"), 200) + "\n")     # warnings of a good run stay visible
    with open(os.path.join(log_dir, "rank0.stdout"), "rb") as fh:
        sys.stdout.write(fh.read().decode("utf-8", "replace"))
    sys.stdout.flush()
    return 0


class _stdout_to_stderr:
    """RCCL and gloo print a banner on STDOUT when a communicator comes up; stdout is reserved for the one
    JSON line, so fd 1 points at stderr while the process group is created."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def selftest_rank():
    """--selftest: the distributed plumbing of a step without a GPU (gloo): shard bounds, padded
    all-gather, max-over-ranks reduction, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    from speaker_verification_amd import distributed as svdist
    from datetime import timedelta
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if os.environ.get("SVK_BENCH_FAIL_RANK") == str(rank):      # test hook: this rank dies before the rendezvous
        sys.stderr.write("rank %d: SVK_BENCH_FAIL_RANK set, exiting with code 3 before init_process_group\n" % rank)
        return 3
    if world > 1:
        with _stdout_to_stderr():
            dist.init_process_group("gloo", timeout=timedelta(seconds=INIT_TIMEOUT_S))
            dist.barrier()
    n_total = 1003
    lo, hi = svdist.shard_bounds(n_total, world, rank)
    local = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 4)
    full = svdist.all_gather_embeddings(local, n_total)
    ok = bool(torch.equal(full[:, 0], torch.arange(n_total, dtype=torch.float32)))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"selftest": True, "ranks_seen": dist.get_world_size() if world > 1 else 1,
                          "backend": dist.get_backend() if world > 1 else None, "gathered_ok": ok,
                          "max_over_ranks": float(t.item()), "n_gpus": world}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


# --------------------------------------------------------------------------------------------------
# side benches of the hand-written kernels
# --------------------------------------------------------------------------------------------------
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
        doc = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None, None
    rec = doc.get(kernel) or doc.get(kernel + "<merged>") or {}
    return rec.get("hbm_traffic_bytes_per_launch"), os.path.basename(files[-1])


def pmc_counter(kernel, counter, per=1024.0):
    """`counter` per cube of `kernel` from the newest committed PMC summary (the c3d2_* kernels were profiled over
    `bench.py --c3d2-only`: 1 024 cubes per launch), or None.  (Round 3's summaries call the first block
    "c3d2_stage1w_kernel<merged>": the same kernel before the other forms were pruned.)"""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_frontend_pmc.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        rec = doc.get(kernel) or doc.get(kernel + "<merged>") or {}
        if counter in rec:
            return rec[counter] / per, os.path.basename(path)
    return None, None


def pmc_provenance(name):
    """(what profiles/<name> says it was collected from, whether that differs from the kernel sources this run uses)."""
    from speaker_verification_amd import _lib
    now = _lib.provenance()
    try:
        was = json.load(open(os.path.join(REPO, "profiles", name))).get("_provenance") or {}
    except (OSError, ValueError, TypeError):
        was = {}
    return was, now, was.get("csrc_sha") != now["csrc_sha"]


def committed_one_gpu_value(world):
    """{"one_gpu_value", "n_times_one_gpu_value", "source"} from the newest committed 1-GPU bench line under profiles/."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_bench.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("n_gpus") == 1 and rec.get("value"):
            return {"one_gpu_value": rec["value"], "n_times_one_gpu_value": world * rec["value"], "source": os.path.basename(path)}
    return None


def _median_ms(torch, fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


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
    t = _median_ms(torch, lambda: eng.features(pcm, spec), reps) * 1e-3
    bytes_per_utt = 48000 * 2 + 297 * 40 * 4
    gbs = n_clips * bytes_per_utt / t / 1e9
    traffic, src = pmc_traffic("frontend_kernel<int16,nfft1024>")
    return {"workload": "1024 x 3 s clips, pre-emph + lmfe-40 (nfft 1024)", "utt_per_s": n_clips / t,
            "frontend_kernel_ms": t * 1e3,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                         "algorithmic_bytes_per_launch": n_clips * bytes_per_utt,
                         "kernel": "frontend_kernel<int16,nfft1024>", "bytes_per_utt": bytes_per_utt}}


def stage_kernels_bench(eng, torch, reps=10):
    """The stage-level kernels behind the individual speechpy.processing functions, each on
    its own: HIP-event ms, algorithmic HBM bytes (reads + writes of the arrays the function is defined
    on), GB/s and the fraction of the 8 TB/s roof."""
    from speaker_verification_amd import _lib
    from speaker_verification_amd.speechpy import feature as ffeat
    g = torch.Generator(device=eng.device)
    g.manual_seed(5)
    rows = {}

    def add(name, fn, nbytes, note):
        ms = _median_ms(torch, fn, reps)
        gbs = nbytes / ms / 1e6
        rows[name] = {"ms": ms, "algorithmic_bytes": int(nbytes), "GBps": gbs, "frac": gbs / HBM_PEAK_GBS, "workload": note}

    # cmvnw: 1 024 clips x 298 frames x 13 cepstra, window 301 (processing.py:274-327), with variance
    x = torch.randn((1024, 298, 13), device=eng.device, generator=g)
    add("cmvnw(win 301, variance)", lambda: eng.cmvnw(x, 301, True), 2 * 2 * x.numel() * 4,
        "1024 x (298, 13) f32, two passes (mean, then std of the centred rows): each reads + writes the array")
    xl = torch.randn((64, 6000, 40), device=eng.device, generator=g)
    add("cmvnw(win 301) long", lambda: eng.cmvnw(xl, 301, False), 2 * xl.numel() * 4, "64 x (6000, 40) f32 (60 s clips)")
    # spectrum: 32 768 frames
    nfr = 32768
    for nfft, flen in ((512, 320), (1024, 400), (256, 200), (2048, 400), (400, 400)):
        fr = torch.randn((nfr, flen), device=eng.device, generator=g)
        add("power_spectrum(nfft %d)" % nfft, lambda fr=fr, nfft=nfft: eng.spectrum(fr, nfft, True),
            nfr * (flen + nfft // 2 + 1) * 4, "%d frames of %d samples -> %d bins" % (nfr, flen, nfft // 2 + 1))
    # mel / log / DCT stage on a power spectrum
    for bins, nf in ((1025, 80), (129, 40)):
        p = torch.rand((nfr, bins), device=eng.device, generator=g) + 0.1
        bank = eng.to_device(ffeat.filterbanks(nf, bins, 16000, 0, 8000), torch.float32)
        add("mel_features(%d bins, %d filters, mfcc-13)" % (bins, nf),
            lambda p=p, bank=bank: eng.mel_features(p, bank, _lib.OUT_MFCC, 13, True, False),
            nfr * (bins + 13) * 4, "%d frames" % nfr)
    sig = torch.randn((4800000,), device=eng.device, generator=g)
    add("preemphasis", lambda: eng.preemphasis(sig), sig.numel() * 8, "4.8 M f32 samples")
    add("stack_frames(400/160)", lambda: eng.stack_frames(sig, 400, 160, 29997), sig.numel() * 4 + 29997 * 400 * 4,
        "4.8 M samples -> 29 997 frames")
    feat = torch.randn((1024 * 298, 40), device=eng.device, generator=g)
    add("derivative_extraction", lambda: eng.derivative(feat, 2), 2 * feat.numel() * 4, "305 152 x 40 f32")
    pw = torch.rand((nfr, 257), device=eng.device, generator=g)
    add("log_power_spectrum(normalize)", lambda: eng.log_power_(pw.clone(), True), 3 * pw.numel() * 4,
        "32 768 x 257 f32: log pass (read + write) + subtract-max pass; timed with the clone (1 read + 1 write more)")
    return rows


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
    t, (vlen, gather) = timed(lambda: pipe.vad(chunk))
    kept = float(vlen.float().sum().item()) if vlen is not None else float(n * L)
    rows["vad (frame flags + hysteresis + index of the kept frames; no copy)"] = (t, n * L * 2)
    t, (feat, nf, _) = timed(lambda: eng.features(chunk, pipe.spec, lengths=vlen, gather=gather))
    frames = float(nf.float().sum().item())
    rows["frontend (lmfe-40, nfft 1024)"] = (t, kept * 2 + frames * 40 * 4)
    t, _ = timed(lambda: eng.cmvn_(feat, nf, variance=True))
    rows["cmvn"] = (t, 2 * frames * 40 * 4)
    t, idx = timed(lambda: eng.draw_crops(nf, c.CUBE_CROPS, c.CUBE_FRAMES, pipe.crop_seed, first_utt, pipe.bad_clips))
    rows["draw_crops"] = (t, n * (4 + 80))
    t, emb = timed(lambda: pipe.embed_features(feat, idx))
    rows["cube + C3D2 forward"] = (t, None)
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
    # (the chip raises its clock over the first dozens of launches of a cold kernel: measured 0.52 -> 0.44 ms for
    # the same shape depending only on how many launches came before; warm up generously, then take the median)
    ms = _median_ms(torch, lambda: eng.cosine_scores(t, e), max(reps, 30), warm=20)
    tf = 2.0 * nt * ns * d / ms / 1e9
    return {"workload": "%d x %d x %d cosine score matrix" % (nt, ns, d), "ms": ms,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / F32_MATRIX_PEAK_TFLOPS,
                         "traffic": None, "kernel": "cosine_tiled_kernel<true> (v_mfma_f32_16x16x4_f32)",
                         "output_GBps": nt * ns * 4 / ms / 1e6}}


def ingest_bench(eng, torch, reps=10, n_clips=1024):
    """svk_ingest_resample: 1 024 x 3 s mono clips at 48 kHz -> 16 kHz float32.
    Algorithmic bytes per clip: 144 000 x 2 read + 48 000 x 4 written = 480 000 B."""
    from speaker_verification_amd import ingest
    pcm = (torch.randn(n_clips, 144000, device=eng.device) * 3000).to(torch.int16)
    up, down = ingest.rational_ratio(48000, 16000)
    taps = ingest.resample_taps(up, down)
    ms = _median_ms(torch, lambda: eng.resample(pcm, up, down, taps), reps)
    gbs = n_clips * 480000 / ms / 1e6
    return {"workload": "%d x 3 s clips, 48 kHz mono int16 -> 16 kHz float32" % n_clips, "ms": ms,
            "utt_per_s": n_clips / ms * 1e3,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": None, "kernel": "decimate_kernel<3, float>",
                         "bytes_per_utt": 480000}}


def ragged_lengths(n, seed=7):
    """VoxCeleb1-like utterance lengths in samples (README.md:4-7 / SURVEY 8f-2: 4 .. 145 s, median ~ 7-8 s):
    4 s + a log-normal tail, clipped at 145 s, seeded."""
    rng = np.random.default_rng(seed)
    sec = np.minimum(4.0 + rng.lognormal(mean=np.log(3.2), sigma=0.85, size=n), 145.0)
    sec[0], sec[1] = 145.0, 4.0                                  # both ends of the range are always present
    return (sec * 16000).astype(np.int64)


def ragged_bench(pipe, eng, torch, n_clips=2048, reps=3):
    """The realistic-length workload (/root/reference/load_data.py:23-53 reads whole VoxCeleb files,
    utils.py:170-173): `n_clips` clips of 4 .. 145 s through `embed_ragged_resident` (audio already in HBM, addressed by
    offsets) and `embed_ragged` (a list of host arrays: packed per batch and uploaded).  The network costs the same per
    clip whatever its length (20 crops of 80 frames); VAD + front end + CMVN scale with the audio -- their share of the
    step is reported from HIP events."""
    from speaker_verification_amd import synth
    dev = eng.device
    lens = ragged_lengths(n_clips)
    slots = (lens + 7) // 8 * 8
    offs = np.concatenate([[0], np.cumsum(slots)[:-1]]).astype(np.int64)
    total = int(slots.sum())
    # clip k = consecutive 3 s device clips of ONE synthetic speaker, cut to its length (speech bursts with gaps throughout)
    base, _ = synth.corpus_device(1024, dev, first_clip=0, utts_per_speaker=UTTS_PER_SPK)
    flat = base.reshape(-1)
    buf = torch.zeros((total,), dtype=torch.int16, device=dev)
    rng = np.random.default_rng(11)
    for k in range(n_clips):
        start = int(rng.integers(0, 1024 - 49)) * synth.CLIP_SAMPLES     # 49 segments cover 145 s
        buf[offs[k]:offs[k] + lens[k]] = flat[start:start + int(lens[k])]
    del base, flat
    saved_mb = pipe.micro_batch
    pipe.micro_batch = 1024
    try:
        emb = pipe.embed_ragged_resident(buf, offs, lens)                 # warm
        torch.cuda.synchronize()
        times, shares = [], []
        for _ in range(reps):
            spans = []
            t0 = time.perf_counter()
            emb = pipe.embed_ragged_resident(buf, offs, lens, spans=spans)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            by = {}
            for name, a, b in spans:
                by[name] = by.get(name, 0.0) + a.elapsed_time(b)
            shares.append(by)
        t_res = float(np.median(times))
        stage_ms = {k: float(np.median([x.get(k, 0.0) for x in shares])) for k in ("vad", "frontend", "cmvn", "crops", "gather", "network")}
        front_ms = sum(v for k, v in stage_ms.items() if k != "network")
        net_ms = stage_ms["network"]
        n_batches = sum(1 for name, _, _ in spans if name == "frontend")
        host_buf = buf.cpu().numpy()
        clips = [host_buf[offs[k]:offs[k] + lens[k]] for k in range(n_clips)]
        pipe.embed_ragged(clips)                  # warm: the pinned / device staging buffers are sized by the largest batch
        torch.cuda.synchronize()
        th = []
        for _ in range(max(3, reps)):             # (host packing competes with whatever else the box runs: a median, not one shot)
            t0 = time.perf_counter()
            emb_h = pipe.embed_ragged(clips)
            torch.cuda.synchronize()
            th.append(time.perf_counter() - t0)
        t_host = float(np.median(th))
        same = float((emb_h - emb).abs().max().item())
        tp = []
        for _ in range(max(3, reps)):             # the same clips handed over as ONE host arena + offsets (a loader that decodes into one buffer)
            t0 = time.perf_counter()
            emb_p = pipe.embed_ragged_resident(host_buf, offs, lens)
            torch.cuda.synchronize()
            tp.append(time.perf_counter() - t0)
        t_packed = float(np.median(tp))
        same = max(same, float((emb_p - emb).abs().max().item()))
    finally:
        pipe.micro_batch = saved_mb
    audio_s = float(lens.sum()) / 16000.0
    return {"workload": "%d clips of 4 .. 145 s (seeded log-normal lengths: median %.1f s, mean %.1f s, %.0f s of audio, %.0f MB "
                        "of int16 PCM), energy VAD -> lmfe -> CMVN -> crops -> C3D2, batches sorted by length"
                        % (n_clips, float(np.median(lens)) / 16000, audio_s / n_clips, audio_s, total * 2 / 1e6),
            "resident": {"utt_per_s": n_clips / t_res, "audio_seconds_per_s": audio_s / t_res, "ms": t_res * 1e3,
                         "front_end_ms": front_ms, "network_ms": net_ms, "stage_ms": stage_ms, "batches": n_batches,
                         "front_end_share": front_ms / max(front_ms + net_ms, 1e-9)},
            "host_fed": {"utt_per_s": n_clips / t_host, "audio_seconds_per_s": audio_s / t_host, "ms": t_host * 1e3,
                         "note": "a list of host NumPy clips: 8 host threads pack batch k + 1 into a pinned buffer and a side stream uploads it "
                                 "while the GPU works on batch k; same kernels"},
            "host_packed": {"utt_per_s": n_clips / t_packed, "audio_seconds_per_s": audio_s / t_packed, "ms": t_packed * 1e3,
                            "note": "one pageable host array holding every clip + offsets (embed_ragged_resident takes host memory too): "
                                    "a single upload, no per-clip packing on the host; same kernels"},
            "max_abs_diff_host_vs_resident": same, "short_clips": int(pipe.bad_clips.item()),
            "fixed_3s_front_end_share_for_comparison": None}


# --------------------------------------------------------------------------------------------------
# CPU baseline: the oracle ("port") in a fresh child process that never touches the GPU
# --------------------------------------------------------------------------------------------------
_CPU = {}


def _cpu_init(sample_dir, flags):
    import torch
    torch.set_num_threads(1)
    _CPU["pcm"] = np.load(os.path.join(sample_dir, "pcm.npy"), mmap_mode="r")
    _CPU["crops"] = np.load(os.path.join(sample_dir, "crops.npy"))
    _CPU["state"] = torch.load(os.path.join(sample_dir, "state.pt"), map_location="cpu", weights_only=True)
    _CPU["flags"] = flags


def _cpu_features(i):
    """vad -> /32768 (what librosa hands lmfe) -> preemphasis -> lmfe -> cmvn of sample clip i (one core)."""
    from oracle import speechpy_ref, vad_ref
    from speaker_verification_amd import constants as c
    preemph, cmvn, use_vad = _CPU["flags"]
    clip = np.asarray(_CPU["pcm"][i])
    if use_vad:
        _, _, clip = vad_ref.vad_energy(clip, c.SAMPLE_RATE, c.VAD_FRAME_MS, c.VAD_PADDING_MS, c.VAD_ENERGY_THRESHOLD)
    sig = clip / 32768.0
    sig = speechpy_ref.preemphasis(sig, cof=0.98) if preemph else sig
    feat = speechpy_ref.lmfe(sig, c.SAMPLE_RATE, c.FRAME_LEN, c.FRAME_STEP, c.NUM_COEF, c.NUM_FFT)
    return speechpy_ref.cmvn(feat, variance_normalization=True) if cmvn else feat


def _cpu_chain(i):
    """The reference's per-utterance sequence: features -> cube -> C3D2 at batch 1."""
    from oracle import model_ref
    cube = model_ref.feature_cube(_cpu_features(i), _CPU["crops"][i])[None]
    return model_ref.c3d2_embed(_CPU["state"], cube).numpy()[0]


def _cpu_count_frames(i):
    return _cpu_features(i).shape[0]


def _cpu_worker_ready(counter):
    """Pool initializer: one untimed pass of the whole chain in THIS worker (imports, first-touch of the mapped sample,
    torch-CPU's first convolution), then report in."""
    _cpu_chain(0)
    with counter.get_lock():
        counter.value += 1


def _cpu_score_rows(job):
    """evaluation.py:73-77 for a block of test embeddings: one sklearn-style call per (utterance, speaker) pair."""
    from oracle import scoring_ref
    rows, enroll = job
    return [scoring_ref.compute_similarity(r, enroll) for r in rows]


def cpu_child(sample_dir):
    """Runs with OMP/MKL/OPENBLAS_NUM_THREADS=1 in the environment (set by the parent before this
    interpreter started, i.e. before NumPy was imported).  Prints one JSON object."""
    import multiprocessing as mp
    import platform

    import scipy
    import torch
    from oracle import model_ref, scoring_ref
    meta = json.load(open(os.path.join(sample_dir, "meta.json")))
    flags = (meta["preemph"], meta["cmvn"], meta["vad"])
    _cpu_init(sample_dir, flags)
    n = _CPU["pcm"].shape[0]
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    # a container may see every hardware thread of the host and still be granted a fraction of them (cgroup CPU quota): pools
    # sized by os.cpu_count() then time-slice, and "utt/s on N workers" says little about N cores
    quota = None
    try:
        parts = open("/sys/fs/cgroup/cpu.max").read().split()
        if parts[0] != "max":
            quota = float(parts[0]) / float(parts[1])
    except (OSError, ValueError, IndexError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    cpu_model = platform.processor() or ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {"cores": cores, "os_cpu_count": os.cpu_count(), "cpu_quota_cores": quota, "cpu_model": cpu_model,
           "versions": {"python": platform.python_version(), "numpy": np.__version__, "scipy": scipy.__version__,
                        "torch": torch.__version__}, "sample_clips": n, "stages": {}}

    def rate(fn, count):
        t0 = time.perf_counter()
        res = fn()
        return count / (time.perf_counter() - t0), res

    # ---- front end (vad -> preemph -> lmfe -> cmvn): 1 core, then a process pool over utterances ----
    n1 = min(n, 384)
    r, _ = rate(lambda: [_cpu_features(i) for i in range(n1)], n1)
    out["stages"]["frontend_1core"] = {"utt_per_s": r, "clips": n1, "cores": 1}
    ctx = mp.get_context("fork")               # this process has never touched the GPU: fork is safe
    n_par = max(1, int(round(quota))) if quota else cores     # "all cores" = what the container is granted
    with ctx.Pool(n_par) as pool:
        pool.map(_cpu_count_frames, range(min(n, 2 * n_par)), chunksize=1)            # warm the workers
        r, _ = rate(lambda: pool.map(_cpu_count_frames, range(n), chunksize=max(1, n // (8 * n_par))), n)
        out["stages"]["frontend_allcores"] = {"utt_per_s": r, "clips": n, "cores": n_par, "how": "multiprocessing.Pool"}
    # ---- the whole per-utterance chain (batch 1, like evaluation.py:113-121), one single-threaded worker per slot, at
    # several pool sizes: with SMT and memory-bound batch-1 convolutions "one worker per hardware thread" is not the
    # fastest use of the host (1.6 utt/s per thread at 256 workers vs 115 on one).  `value` = the best. ----
    sweep = {max(1, cores // 4), max(1, cores // 2), cores}
    if quota:                       # the granted share of the host: pools of that many workers, and of twice as many
        sweep = {w for w in sweep if w <= 4 * quota} | {max(1, int(round(quota))), max(1, int(round(2 * quota)))}
    sweep = sorted(sweep)
    out["chain_sweep"] = {}
    best = None
    for workers in sweep:
        ready = ctx.Value("i", 0)
        with ctx.Pool(workers, initializer=_cpu_worker_ready, initargs=(ready,)) as pool:
            t_w = time.perf_counter()
            while ready.value < workers and time.perf_counter() - t_w < 120:          # every worker has run the chain once
                time.sleep(0.01)
            t_ready = time.perf_counter() - t_w
            t0 = time.perf_counter()
            embs_w = np.stack(pool.map(_cpu_chain, range(n), chunksize=max(1, n // (8 * workers))))
            t_chain = time.perf_counter() - t0
            # pair-by-pair scoring (evaluation.py:73-77), spread over the same workers
            enroll = embs_w[::max(1, n // 40)][:40]
            rows_per_job = max(1, n // (4 * workers))
            t0 = time.perf_counter()
            pool.map(_cpu_score_rows, [(embs_w[lo:lo + rows_per_job], enroll) for lo in range(0, n, rows_per_job)], chunksize=1)
            t_score = time.perf_counter() - t0
        dt_w = t_chain + t_score
        out["chain_sweep"][str(workers)] = {"utt_per_s": n / dt_w, "seconds": dt_w, "workers": workers,
                                            "chain_utt_per_s": n / t_chain, "chain_s": t_chain, "scoring_s": t_score,
                                            "warm_s": t_ready}
        if best is None or n / dt_w > best[0]:
            best = (n / dt_w, dt_w, workers)
        embs = embs_w                                                                # (identical whatever the pool size)
    out["value"], out["seconds"], out["workers"] = best
    np.save(os.path.join(sample_dir, "cpu_emb.npy"), embs)
    # ---- C3D2 alone: batch 1 (evaluation.py:113-121) and batch 64, 1 thread and all threads ----
    cubes = np.stack([model_ref.feature_cube(_cpu_features(i % n), _CPU["crops"][i % n]) for i in range(64)])

    def bounded(fn, per_call, budget=4.0, most=64):
        """utt/s of repeated `fn()` calls: one untimed call, then as many as fit `budget` seconds (torch-CPU
        with every hardware thread on a batch of one can take seconds per call)."""
        fn()
        t0 = time.perf_counter()
        calls = 0
        while calls < most and (calls == 0 or time.perf_counter() - t0 < budget):
            fn()
            calls += 1
        return per_call * calls / (time.perf_counter() - t0), per_call * calls

    for threads in (1, n_par):
        torch.set_num_threads(threads)
        tag = "1thread" if threads == 1 else "allthreads"
        k = [0]

        def one():
            k[0] += 1
            return model_ref.c3d2_embed(_CPU["state"], cubes[k[0] % 64][None])
        r, cnt = bounded(one, 1)
        out["stages"]["c3d2_batch1_" + tag] = {"utt_per_s": r, "clips": cnt, "threads": threads}
        r, cnt = bounded(lambda: model_ref.c3d2_embed(_CPU["state"], cubes), 64, most=8)
        out["stages"]["c3d2_batch64_" + tag] = {"utt_per_s": r, "clips": cnt, "threads": threads}
    print(json.dumps(out))
    return 0


def run_cpu_baseline(sample_pcm, crops, state, preemph, cmvn, use_vad):
    """Write the sample, run `bench.py --cpu-child` in a fresh interpreter pinned to one BLAS/OpenMP thread
    per process, read its JSON and the embeddings it computed."""
    import tempfile

    import torch
    d = tempfile.mkdtemp(prefix="svk_cpu_")
    np.save(os.path.join(d, "pcm.npy"), sample_pcm)
    np.save(os.path.join(d, "crops.npy"), crops)
    torch.save(state, os.path.join(d, "state.pt"))
    json.dump({"preemph": preemph, "cmvn": cmvn, "vad": use_vad}, open(os.path.join(d, "meta.json"), "w"))
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1",
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    t0 = time.perf_counter()
    proc = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", d], env=env,
                          stdout=subprocess.PIPE, timeout=900)
    wall = time.perf_counter() - t0
    if proc.returncode != 0:
        raise RuntimeError("cpu baseline child failed with code %d" % proc.returncode)
    rec = json.loads(proc.stdout.decode().strip().splitlines()[-1])
    rec["wall_s"] = wall
    emb = np.load(os.path.join(d, "cpu_emb.npy"))
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    return rec, emb


# --------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.cpu_child:
        return cpu_child(args.cpu_child)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])             # before torch / any GPU call in this process
    if args.selftest:
        return selftest_rank()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # (before torch: the host driver of this pool only supports dmabuf IPC; RCCL's intra-node transport needs this)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from datetime import timedelta
    n_dev = torch.cuda.device_count()                      # (counting devices does not initialise the GPU)
    if os.environ.get("SVK_BENCH_FAIL_RANK") == str(rank):
        sys.stderr.write("rank %d: SVK_BENCH_FAIL_RANK set, exiting with code 3 before init_process_group\n" % rank)
        return 3
    if args.backend == "nccl" and world > n_dev:
        # RCCL wants one device per rank: say so before any GPU call instead of failing inside the rendezvous
        sys.stderr.write("bench.py: --gpus %d over RCCL needs %d devices, this host shows %d (torch.cuda.device_count()); "
                         "use --backend gloo to rehearse several ranks on one GPU\n" % (world, world, n_dev))
        return 2
    device_index = local_rank if args.backend == "nccl" else local_rank % max(1, n_dev)
    torch.cuda.set_device(device_index)
    use_dist = world > 1 or os.environ.get("SVK_BENCH_FORCE_DIST") == "1"   # the latter: exercise RCCL with one rank
    if use_dist:
        with _stdout_to_stderr():
            # a rank that died before the rendezvous must not hold the others for torch's default 10 / 30 minutes
            # (the launcher ends them within seconds; under torch.distributed.run this timeout is the bound)
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", device_index),
                                        timeout=timedelta(seconds=INIT_TIMEOUT_S))
            else:
                dist.init_process_group("gloo", timeout=timedelta(seconds=INIT_TIMEOUT_S))
            dist.barrier()
            torch.cuda.synchronize()

    from speaker_verification_amd import constants as c, distributed as svdist, evaluation, synth
    from speaker_verification_amd.engine import get_engine
    from speaker_verification_amd.model import C3D2, calibrate_batchnorm, seeded_model
    from speaker_verification_amd.pipeline import VerificationPipeline, enroll_last_utterance

    eng = get_engine(device_index)
    dev = eng.device

    if args.frontend_only:
        res = {"A": frontend_A_bench(eng, torch, reps=max(args.steps, 5)),
               "B": frontend_B_bench(eng, torch, reps=max(args.steps, 5)),
               "ingest_resample": ingest_bench(eng, torch)}
        print(json.dumps(res))
        return 0
    if args.stages_only:
        print(json.dumps({"stage_kernels": stage_kernels_bench(eng, torch, reps=max(args.steps, 5)),
                          "cosine_mfma": cosine_mfma_bench(eng, torch)}))
        return 0

    def bench_model():
        """The committed trained checkpoint (tools/train_synth_checkpoint.py: 100 synthetic speakers disjoint from the
        corpus's, reference format, loaded weights-only), or with --random-init the seeded random-init network of rounds 1-3."""
        if args.random_init:
            return seeded_model(2024, n_labels=1211), {"weights": "random-init (seed 2024), BatchNorm calibrated on 256 clips"}
        ck = torch.load(args.checkpoint, map_location="cpu", weights_only=True)
        m = C3D2(int(ck["state_dict"]["FC6.weight"].shape[0]), 1)
        m.load_state_dict(ck["state_dict"])
        meta = {k: ck.get("meta", {}).get(k) for k in ("tool", "speakers", "utts_per_speaker", "steps", "init_seed", "clip_seed",
                                                        "held_out_eer")}
        meta["weights"] = os.path.relpath(args.checkpoint, REPO)
        return m.eval(), meta

    if args.ragged_only:
        pipe = VerificationPipeline(bench_model()[0], use_vad=not args.no_vad, normalize=not args.no_cmvn,
                                    preemph_cof=None if args.no_preemph else 0.98, crop_rng="device", micro_batch=1024)
        print(json.dumps(ragged_bench(pipe, eng, torch)))
        return 0
    if args.c3d2_only:
        pcm, _ = synth.corpus_device(1024, dev, first_clip=0, utts_per_speaker=UTTS_PER_SPK)
        pipe = VerificationPipeline(bench_model()[0], use_vad=not args.no_vad, normalize=not args.no_cmvn,
                                    preemph_cof=None if args.no_preemph else 0.98, crop_rng="device", micro_batch=1024)
        vlen, gather = pipe.vad(pcm)
        feat, n_frames = pipe.features(pcm, vlen, gather)
        idx = eng.draw_crops(n_frames, c.CUBE_CROPS, c.CUBE_FRAMES, pipe.crop_seed, 0, pipe.bad_clips)
        ms = _median_ms(torch, lambda: pipe.embed_features(feat, idx), max(args.steps, 5))
        print(json.dumps({"workload": "1024 cubes: svk_c3d2_stage1 + stage2 + conv31 + conv32t + conv41 + conv42 + fc5",
                          "ms": ms, "utt_per_s": 1024 / ms * 1e3, "tflops": 1024 * C3D2_GFLOP_PER_UTT / ms}))
        return 0
    n_total = args.corpus
    lo_r, hi_r = svdist.shard_bounds(n_total, world, rank)
    n_local = hi_r - lo_r
    pcm, _ = synth.corpus_device(n_local, dev, first_clip=lo_r, utts_per_speaker=UTTS_PER_SPK)
    model, weights_meta = bench_model()
    if args.micro_batch <= 0:
        # live per clip while the second block runs: 663 552 (after pool1) + 903 168 (conv2_1's scratch) + 161 280 (after pool2)
        # + 47 520 (features) + the VAD's index and slack: ~2 MB.  The corpus is already resident: mem_get_info sees what is left
        # (a whole number of equal sequences: 148 642 clips on one 288 GB card = 2 x 74 321, and still 2 with 20 GB less free)
        try:
            free_bytes, _ = torch.cuda.mem_get_info(dev)
            free_bytes /= max(1, -(-world // max(1, torch.cuda.device_count())))     # (ranks rehearsing on one card share it)
            sequences = max(1, int(np.ceil(n_local * 2.0e6 / (0.6 * free_bytes))))
            args.micro_batch = max(1024, -(-n_local // sequences))
        except RuntimeError:
            args.micro_batch = 4096
    pipe = VerificationPipeline(model, use_vad=not args.no_vad, normalize=not args.no_cmvn,
                                preemph_cof=None if args.no_preemph else 0.98, crop_rng="device",
                                micro_batch=args.micro_batch,
                                overlap_front=os.environ.get("SVK_BENCH_OVERLAP", "0") == "1")
    if args.random_init:
        # random-init weights with BatchNorm statistics calibrated on the first 256 clips of the corpus, identically on
        # every rank (model.calibrate_batchnorm explains why).  The statistics come from a training-mode forward of the
        # torch module: run on the HOST (torch-CPU, a few seconds, outside the timed region) with this rank's share of the
        # host's threads (8 ranks x 32 threads at once would oversubscribe it)
        cal_pcm, _ = synth.corpus_device(256, dev, first_clip=0, utts_per_speaker=UTTS_PER_SPK)
        _, cal_cubes = pipe.crops_and_cubes(cal_pcm)
        threads = torch.get_num_threads()
        torch.set_num_threads(max(1, min(32, threads, (os.cpu_count() or 1) // world)))
        cpu_model = seeded_model(2024, n_labels=1211)
        calibrate_batchnorm(cpu_model, cal_cubes.cpu())
        torch.set_num_threads(threads)
        pipe.model.load_state_dict(cpu_model.state_dict())
        pipe.refresh_model()
        del cal_pcm, cal_cubes, cpu_model
    n_test = min(N_TEST, n_total)
    spk_all = (np.arange(n_total) // UTTS_PER_SPK).astype(np.int32)
    ids, last = enroll_last_utterance(None, spk_all[:n_test])                  # Q17: last utterance enrols
    last_dev = torch.from_numpy(last).to(dev)
    spans = pipe.chunks(n_local)

    fe_events, ag_events = [], []

    def one_step(record):
        # timed region: everything from resident PCM to the score matrix
        pipe.kernel_events = [] if record and pipe.kernel_events is None else pipe.kernel_events
        local = torch.empty((n_local, 128), dtype=torch.float32, device=dev)
        # SVK_BENCH_OVERLAP=1: the pipeline's own two-stream form (the front of sequence k + 1 on a side stream under the network
        # of sequence k: VerificationPipeline._embed_overlapped) instead of the loop below -- an experiment switch, off by default
        for lo, hi in ([] if (pipe.overlap_front and len(spans) > 1) else spans):
            chunk = pcm[lo:hi]
            vlen, gather = pipe.vad(chunk)
            if record:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            feat, n_frames, _ = eng.features(chunk, pipe.spec, lengths=vlen, gather=gather)
            if record:
                b.record()
                fe_events.append((a, b))
            if pipe.normalize:
                eng.cmvn_(feat, n_frames, variance=True)
            idx = eng.draw_crops(n_frames, c.CUBE_CROPS, c.CUBE_FRAMES, pipe.crop_seed, lo_r + lo, pipe.bad_clips)
            local[lo:hi] = pipe.embed_features(feat, idx)      # feature rows + crop starts -> the seven network kernels
        if pipe.overlap_front and len(spans) > 1:
            local = pipe.embed(pcm, first_utt=lo_r)
        if record:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        full = svdist.all_gather_embeddings(local, n_total)
        if record:
            b.record()
            ag_events.append((a, b))
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
    per_rank_ms = [dt / args.steps * 1e3]
    if use_dist:
        mine = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                       # each rank's own wall time of the K steps ...
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        dt = max(float(t.item()) for t in every)           # ... and the max over ranks is the job's
    bad = int(pipe.bad_clips.item())

    result = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_total * args.steps / dt
        # roofline of the fused front-end kernel: HIP events on the launch stream around each launch of the
        # timed region; algorithmic bytes of the launches as they ran (clips shortened by the VAD): kept
        # samples x 2 B read + produced frames x 40 x 4 B written, counted in one extra untimed VAD pass
        fe_ms = np.array([a.elapsed_time(b) for a, b in fe_events])
        kept = torch.zeros((), dtype=torch.float64, device=dev)
        frames = torch.zeros((), dtype=torch.float64, device=dev)
        for lo, hi in spans:
            vlen, _ = pipe.vad(pcm[lo:hi])
            vl = vlen.to(torch.float64) if vlen is not None else torch.full((hi - lo,), float(pcm.shape[1]), device=dev,
                                                                           dtype=torch.float64)
            kept += vl.sum()
            frames += torch.clamp(torch.floor((vl - pipe.spec.frame_len) / pipe.spec.frame_stride), min=0).sum()
        launches = len(spans)
        bytes_per_launch = (float(kept.item()) * 2 + float(frames.item()) * 40 * 4) / launches
        avg_launch_s = float(fe_ms.mean()) * 1e-3
        gbs = bytes_per_launch / avg_launch_s / 1e9
        traffic, traffic_src = pmc_traffic("frontend_kernel<int16,nfft1024>")
        kernel_events, pipe.kernel_events = pipe.kernel_events, None   # stop recording: the rest is outside the timed region
        labels = (spk_all[:n_test, None] == ids[None, :]).astype(np.float64)
        sc = scores.cpu().numpy().astype(np.float64)
        eer, auc, _, _ = evaluation.get_eer_auc(labels.flatten(), sc.flatten())
        eer_dev, auc_dev = evaluation.get_eer_auc_device(labels, scores)       # svk_roc_eer on the same matrix
        e2e_tflops = value * C3D2_GFLOP_PER_UTT / 1e3
        frontend_roofline = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                             "traffic_note": "PMC pass ran this kernel on 1 024 FULL 3 s clips per launch "
                                             "(algorithmic 146 964 480 B); `achieved` uses the VAD-shortened bytes of "
                                             "the launches timed here",
                             "kernel": "frontend_kernel<int16,nfft1024>", "avg_launch_ms": avg_launch_s * 1e3,
                             "launches_per_step": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
                             "full_clip_bytes_per_utt": 48000 * 2 + 297 * 40 * 4,
                             "share_of_step": float(fe_ms.sum()) / args.steps / ms_per_step}
        main_roofline = frontend_roofline
        stage2_roofline, network_rows, issued_gflop_per_utt = None, None, None
        if kernel_events:
            # Every network kernel, from HIP events on the launch stream around each launch of the timed region.
            #   frac            = ISSUED matrix work / time / peak: MFMA wave-instructions per cube (the committed SQ_INSTS_MFMA
            #                     counter; by construction when no profile has the kernel) x 2 048 FLOP x cubes -- a share of
            #                     the matrix pipe's issue slots, <= 1 by definition;
            #   algorithmic_frac = SURVEY 8(d)'s direct-form multiply-adds x 2 / time / peak: what the reference's sums would
            #                     cost -- it exceeds `frac` (and may pass 1) where Winograd F(2,3) along depth issues 2/3 of them.
            cubes_total = float(sum(sp["cubes"] for sp in kernel_events))
            network_rows = {}
            issued_gflop_per_utt = 0.0
            pipe_seconds_per_utt = 0.0       # time the matrix pipes need for the issued work at their peaks (f32 and f16 rows each at its own)
            covered_ms = 0.0
            for name, symbols, mmac, mfma_design, mfma_flop, peak in NETWORK_KERNELS:
                evs = [sp[name] for sp in kernel_events if name in sp]
                if not evs:
                    continue
                ms = float(sum(a.elapsed_time(b) for a, b in evs))
                mfma, src = 0.0, []
                for sym in symbols:
                    got, where = pmc_counter(sym, "SQ_INSTS_MFMA")
                    src.append(where)
                    mfma += got if got is not None else 0.0
                if not all(src):
                    mfma, src = float(mfma_design), ["by construction (no committed counter)"]
                elif pmc_provenance(src[0])[2] and abs(mfma - mfma_design) > 1e-3 * mfma_design:
                    # counters of OTHER kernel sources that no longer state this code's MFMA count: the count by construction is
                    # what ran (the row stays marked stale; the vector-instruction ratio below still describes the profiled code)
                    mfma = float(mfma_design)
                # non-MFMA vector instructions per MFMA (committed SQ_INSTS_VALU, which counts the MFMAs too): f32 MFMA and f32
                # VALU never run together on this chip (SQ_VALU_MFMA_COEXEC_CYCLES = 0 in every profile), so a VALU
                # wave-instruction (4 cycles of a SIMD where the MFMA takes 32) is paid in the same issue slots
                valu = 0.0
                for sym in symbols:
                    got, _ = pmc_counter(sym, "SQ_INSTS_VALU")
                    valu = None if (got is None or valu is None) else valu + got
                valu_per_mfma = None if (valu is None or not all(src) or mfma <= 0) else (valu - mfma) / mfma
                tf_issued = cubes_total * mfma * mfma_flop / ms / 1e9
                tf_alg = cubes_total * 2 * mmac * 1e6 / ms / 1e9
                issued_gflop_per_utt += mfma * mfma_flop / 1e9
                pipe_seconds_per_utt += mfma * mfma_flop / (peak * 1e12)
                covered_ms += ms
                f16 = mfma_flop == F16_MFMA_FLOP
                network_rows[name] = {
                    "bound": "mfma", "achieved": tf_issued, "peak": peak, "unit": "TFLOP/s",
                    "pipe": "f16 (v_mfma_f32_16x16x32_f16, two-piece products: 3 issued products per f32 product)" if f16
                            else "f32 (v_mfma_f32_16x16x4_f32)",
                    "frac": tf_issued / peak, "algorithmic_tflops": tf_alg,
                    # SURVEY 8(d)'s direct-form f32 multiply-adds against the F32 matrix peak, whatever pipe the kernel uses
                    "algorithmic_frac": tf_alg / F32_MATRIX_PEAK_TFLOPS, "kernel": " + ".join(symbols),
                    "avg_launch_ms": ms / len(evs), "cubes_per_launch": cubes_total / len(evs),
                    "mfma_per_cube": mfma, "mfma_per_cube_by_construction": mfma_design, "mfma_source": src[0],
                    "counters_from": None if not all(src) or src[0].startswith("by ") else pmc_provenance(src[0])[0] or "no provenance recorded",
                    "stale": None if not all(src) or src[0].startswith("by ") else pmc_provenance(src[0])[2],
                    "direct_form_mmac_per_cube": mmac, "share_of_step": ms / args.steps / ms_per_step,
                    "valu_per_mfma": valu_per_mfma,
                    # (the f32 MFMA excludes every other vector instruction for its 32 cycles; the f16 one holds the vector issue
                    # for 8 of its 16: no such sum for the first block)
                    "fp32_lanes_busy": None if valu_per_mfma is None or f16
                    else tf_issued / peak * (1.0 + valu_per_mfma * 4.0 / 32.0)}
            r1 = network_rows["stage1"]
            t1, t1_src = pmc_traffic(r1["kernel"])
            main_roofline = dict(r1)
            main_roofline.update({
                "traffic": None if t1 is None else t1 * r1["cubes_per_launch"] / 1024.0, "traffic_source": t1_src,
                "traffic_note": "PMC pass: 1 024 cubes per launch, scaled by cubes_per_launch / 1024; algorithmic bytes per cube: "
                                "47 520 (features, re-read 36 x from L2) + 663 552 written",
                "kernel": r1["kernel"] + " (cube + conv1_1 + conv1_2 + pool1 through two-piece f16 products, "
                          "v_mfma_f32_16x16x32_f16)",
                "note": "frac = issued MFMA work (SQ_INSTS_MFMA per cube x 16 384 FLOP x cubes per launch) / HIP-event launch "
                        "time / the dense f16 matrix peak (16 x the f32 one: MI355X_MICROARCH.md '~2.5 PF'): the share of the F16 "
                        "matrix pipe's issue slots.  Every f32 product is issued as three f16 products (x = h + l: h_x h_w + "
                        "l_x h_w + h_x l_w, f32 accumulation; conv1_1 as two K = 32 blocks of which 3/4 carry products).  "
                        "algorithmic_frac counts SURVEY 8(d)'s direct-form f32 multiply-adds (conv1_1 12.44 M + conv1_2 143.33 M per "
                        "cube) against the F32 matrix peak: what the same sums would need on the pipe the reference's dtype "
                        "names -- above 1 is the point of the construction.  `stale`: the committed counters were "
                        "collected from other kernel sources than this run's (csrc_sha differs); SQ_INSTS_MFMA is then checked "
                        "against mfma_per_cube_by_construction, valu_per_mfma describes the profiled code",
                "this_run": pmc_provenance("")[1]})
            stage2_roofline = network_rows.get("stage2")
            network_rows["_covered_share_of_step"] = covered_ms / args.steps / ms_per_step
        result = {
            "metric": "utterances/sec (MFCC->embed->cosine)", "value": value, "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # f32 in, f32 out, f32 accumulation everywhere; the first two blocks' PRODUCTS are three f16 piece products per f32 product
            # (22-bit operands: 1e-6 of the activation scale from the f32 kernel it replaces, parity bars unchanged)
            "dtype": "f32 (conv1_1 .. conv4_1: two-piece f16 products, f32 accumulation)",
            "data": "synthetic (seeded formant 'voices', 3 s / 16 kHz int16, generated on device); C3D2 weights: %s"
                    % ("seeded random init" if args.random_init else
                       "the committed checkpoint trained on synthetic speakers disjoint from the corpus's (tools/train_synth_checkpoint.py)"),
            "weights": weights_meta,
            "config": {"workload": "configs[4]: %d-clip corpus over %d rank(s) (%d clips on rank 0): energy-VAD -> pre-emph + "
                                   "lmfe(25ms/10ms/1024/40) -> CMVN -> 20x80x40 cube -> C3D2(f32) -> all-gather -> "
                                   "%dx%d cosine" % (n_total, world, n_local, n_test, len(ids)),
                       "corpus_clips": n_total, "clips_per_rank": n_local, "micro_batch": spans[0][1] - spans[0][0],
                       "vad": pipe.use_vad, "cmvn": pipe.normalize, "preemph": not args.no_preemph,
                       "parallelism": "dp%d" % world, "crop_rng": "device"},
            "ranks_seen": dist.get_world_size() if use_dist else 1,
            "backend": ("rccl (torch 'nccl')" if args.backend == "nccl" else "gloo") if use_dist else None,
            "allgather_us": float(np.median([a.elapsed_time(b) for a, b in ag_events])) * 1e3 if use_dist else None,
            "allgather_bytes_per_rank": svdist.shard_rows(n_total, world) * 128 * 4,
            "per_rank_ms": per_rank_ms,
            "slowest_rank": int(np.argmax(per_rank_ms)), "fastest_rank": int(np.argmin(per_rank_ms)),
            # what N GPUs would do at the committed 1-GPU rate (the driver computes the efficiency from its own per-N runs)
            "scaling_efficiency_vs": committed_one_gpu_value(world),
            # what shaped RCCL's choice of algorithm / transport, when the caller set any of it (NCCL_DEBUG=INFO prints the
            # ring / tree and the xGMI links picked to each rank's stderr file under gpurun_out/bench_ranks/)
            "rccl_env": {k: v for k, v in os.environ.items()
                         if k.startswith(("NCCL_", "RCCL_")) or k in ("HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES")},
            "rccl_expectation_us": {"direct_one_shot": 62, "ring": 435,
                                    "note": "SURVEY section 5, 9.5 MB per rank at 8 ranks over 153 GB/s xGMI links"},
            "roofline": main_roofline,
            "roofline_frontend": frontend_roofline,
            "roofline_stage2": stage2_roofline,
            "roofline_network": network_rows,
            "roofline_e2e": {"bound": "mfma",
                             "achieved": None if issued_gflop_per_utt is None else value * pipe_seconds_per_utt / world,
                             "peak": 1.0, "unit": "matrix-pipe issue time / wall time",
                             "frac": None if issued_gflop_per_utt is None else value * pipe_seconds_per_utt / world,
                             "issued_gflop_per_utt": issued_gflop_per_utt,
                             "algorithmic_tflops": e2e_tflops,
                             "algorithmic_frac": e2e_tflops / (F32_MATRIX_PEAK_TFLOPS * world),
                             "gflop_per_utt": C3D2_GFLOP_PER_UTT,
                             "note": "whole step.  frac = utterances/s x the time the matrix pipes need, at their dense peaks, for the "
                                     "MFMA work the network kernels ISSUE per utterance (sum over the roofline_network rows: "
                                     "conv1_1 .. conv4_1 on the f16 pipe at 16 x the f32 rate, conv4_2 and FC5 on the f32 pipe; with every "
                                     "kernel on the f32 pipe this is rounds 2 - 4's definition); algorithmic_frac = utterances/s x "
                                     "SURVEY 8(d)'s 676.6 MFLOP of direct-form f32 sums / the f32 matrix peak (ceiling 232 k utt/s "
                                     "per GPU on that pipe) -- conv4_2 runs through Winograd F(2,3) along depth and issues 2/3 of "
                                     "its products, conv1_1 .. conv4_1 issue 3 f16 products per f32 product at 16 x the rate"},
            "eer": {"eer": eer, "auc": auc, "eer_device": eer_dev, "auc_device": auc_dev, "pairs": int(labels.size),
                    "short_clips": bad,
                    "note": ("random-init C3D2 (--random-init): an EER near 0.5 is that of an untrained network; the statement is "
                             "numerical only" if args.random_init else
                             "C3D2 trained on 100 synthetic speakers that are not in the corpus (speaker_verification_amd/checkpoints/"
                             "c3d2_synth.json): the 40 test speakers are unseen, the EER is an operating point on a steep ROC")
                            + ".  Host sklearn path = svk_roc_eer = CPU oracle on the same scores; GPU = CPU EER from PCM on the parity sample"},
        }

    if rank == 0 and world == 1 and not args.no_extras and not args.parity_only:
        # the side entries keep round 4's 4 096-clip launch sequences (their numbers stay comparable, and a host-fed run of
        # 8 x 74 321 clips would copy the whole corpus to the host four times)
        pipe.micro_batch = min(pipe.micro_batch, 4096)
        spans = pipe.chunks(n_local)
        # host-fed variant (PCIe included; NOT `value`): 8 micro-batches of the shard from pageable host memory (the
        # first batch's H2D copy is not overlapped)
        n_host = min(n_local, 8 * (spans[0][1] - spans[0][0]))
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
        # the resident path draws crops keyed by the global clip index: the same clips, the same embeddings
        same = pipe.embed(pcm[:n_host])
        result["host_fed"] = {"utt_per_s": n_host / t_h, "utt_per_s_from_pinned": n_host / t_p, "clips": n_host,
                              "max_abs_diff_vs_resident": float(max((emb_h - same).abs().max().item(),
                                                                    (emb_p - same).abs().max().item())),
                              "note": "pageable int16 NumPy, uploaded micro-batch by micro-batch from a helper thread on a copy "
                                      "stream into a device double buffer (no staging copy on the host) -> same kernels; "
                                      "from_pinned: the caller's buffer is pinned"}
        del pinned_pcm, host_pcm
        lo0, hi0 = spans[0]
        result["micro_batch_breakdown"] = stage_breakdown(pipe, eng, torch, pcm[lo0:hi0], 0)
        result["frontend_A"] = frontend_A_bench(eng, torch)
        result["cosine_mfma"] = cosine_mfma_bench(eng, torch)
        result["stage_kernels"] = stage_kernels_bench(eng, torch)
        result["ingest_resample"] = ingest_bench(eng, torch)
        result["ragged"] = ragged_bench(pipe, eng, torch)
        mb = result["micro_batch_breakdown"]["stages"]
        result["ragged"]["fixed_3s_front_end_share_for_comparison"] = float(
            sum(v["share"] for k, v in mb.items() if not k.startswith(("cube", "cosine"))))
    if rank == 0 and world == 1 and (not args.no_extras or args.parity_only):
        if args.cpu_sample > 0:
            # sample: whole speakers from the start of the corpus (the last utterance of each enrols, Q17)
            ns = min(args.cpu_sample, n_local)
            sample = pcm[:ns]
            # the PRODUCTION path (libsvk kernels end to end) on the sample, and the crop starts it drew for the oracle
            crops = pipe.crops_and_cubes(sample, want_cubes=False)
            emb = pipe.embed(sample, crop_idx=crops)
            state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            rec, ref_emb = run_cpu_baseline(sample.cpu().numpy(), crops, state, not args.no_preemph, pipe.normalize,
                                            pipe.use_vad)
            got = emb.cpu().numpy()
            from oracle import scoring_ref
            spk_s = spk_all[:ns]
            uniq, last_s = enroll_last_utterance(None, spk_s)
            lab = (spk_s[:, None] == uniq[None, :]).astype(np.float64)
            s_gpu = pipe.score(emb, emb[torch.from_numpy(last_s).to(dev)]).cpu().numpy().astype(np.float64)
            s_ref = scoring_ref.cosine_matrix(ref_emb, ref_emb[last_s]).astype(np.float64)
            par = {"sample_clips": ns, "embed_max_abs_diff": float(np.abs(got - ref_emb).max()),
                   "embed_scale": float(np.abs(ref_emb).max()),
                   "score_max_abs_diff": float(np.abs(s_gpu - s_ref).max())}
            if lab.shape[1] > 1:
                par["eer_gpu"] = float(evaluation.get_eer_auc(lab.flatten(), s_gpu.flatten())[0])
                par["eer_cpu_ref"] = float(scoring_ref.get_eer_auc(lab.flatten(), s_ref.flatten())[0])
                par["eer_equal"] = par["eer_gpu"] == par["eer_cpu_ref"]
                # how far the two sides' scores are from reordering a pair: the smallest gap between a target score and a
                # non-target score of the CPU side, against the largest GPU - CPU score difference
                tgt, non = np.sort(s_ref[lab > 0]), np.sort(s_ref[lab == 0])
                near = np.abs(non[np.clip(np.searchsorted(non, tgt), 0, non.size - 1)] - tgt)
                near = np.minimum(near, np.abs(non[np.clip(np.searchsorted(non, tgt) - 1, 0, non.size - 1)] - tgt))
                par["min_target_nontarget_gap"] = float(near.min())
            # full-size check: oracle cosine + EER on the SAME embeddings must give the same EER
            fe = full[:n_test].cpu().numpy()
            s_or = scoring_ref.cosine_matrix(fe, fe[last]).astype(np.float64)
            par["full_matrix_score_max_abs_diff"] = float(np.abs(s_or - sc).max())
            par["full_matrix_eer_cpu_ref"] = float(scoring_ref.get_eer_auc(labels.flatten(), s_or.flatten())[0])
            result["parity"] = par
            result["cpu_baseline"] = {
                "value": rec["value"], "unit": "utterances/s", "cores": rec["workers"], "kind": "port",
                "sample": "the first %d clips of the corpus through oracle/ in a fresh process: vad -> /32768 -> preemph "
                          "-> lmfe -> cmvn -> cube -> C3D2 batch 1 -> per-pair cosine, single-threaded workers "
                          "(multiprocessing.Pool; every worker runs the chain once before the clock starts; scoring spread "
                          "over the same workers): the best of %s workers = %d (%.1f s); the host shows %d hardware threads, the "
                          "container's CPU quota is %s"
                          % (ns, sorted(int(k) for k in rec["chain_sweep"]), rec["workers"], rec["seconds"], rec["cores"],
                             "%.1f cores" % rec["cpu_quota_cores"] if rec.get("cpu_quota_cores") else "unlimited"),
                "host_threads": rec["cores"], "cpu_quota_cores": rec.get("cpu_quota_cores"), "chain_by_workers": rec["chain_sweep"],
                "cpu_model": rec["cpu_model"], "os_cpu_count": rec["os_cpu_count"], "versions": rec["versions"],
                "variants": rec["stages"], "child_wall_s": rec["wall_s"]}
    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
