"""End-to-end hot path on one GPU: 16 kHz PCM -> energy VAD -> log-mel front end
(-> CMVN) -> 20 x 80 x 40 feature cube -> C3D2 embedding -> cosine scores.

This is the batched, device-resident form of what the reference does one
utterance at a time on the host:
  vad.py:135-168  ->  load_data.py:50-87  ->  utils.py:351-397  ->
  model.py:141-170  ->  evaluation.py:67-84.
Every stage is a libsvk.so kernel, the C3D2 forward included (`svk_c3d2_stage1`, `svk_c3d2_stage2`,
`svk_c3d2_conv31/32t/41/42`, `svk_c3d2_fc5`: `model.FusedEmbedder`); PyTorch-ROCm serves as device memory, streams and
the module checkpoints load into.
"""
import os

import numpy as np
import torch

from . import _lib
from . import constants as c
from .engine import get_engine, spec_from_seconds


class VerificationPipeline:
    def __init__(self, model, use_vad=True, vad_threshold=c.VAD_ENERGY_THRESHOLD, normalize=c.NORMALIZE,
                 crop_seed=12345, micro_batch=1024, preemph_cof=None, crop_rng="reference", overlap_front=False,
                 pcm_scale=1.0 / 32768.0):
        """model: a `model.C3D2` (one-channel cubes): its inference form is `model.fused_inference()`, seven libsvk kernels.
        crop_rng: "reference" draws crop starts on the host exactly like utils.py:372 (needs the
        per-clip frame counts on the host: one small D2H per micro-batch); "device" draws them
        in a kernel keyed by (crop_seed, global clip index) -- no host round trip.
        preemph_cof: fuse processing.preemphasis(clip, cof=...) in front of the log-mel stage.
        pcm_scale: amplitude factor on the int16 PCM before the front end.  The reference's model path
        reads audio with librosa (utils.py:170-173): float32 in [-1, 1) = int16 / 32768 -> lmfe
        (load_data.py:50-70), and ships NORMALIZE = False, so a reference-format checkpoint expects
        log-mel values of THAT scale (2 ln 32768 = 20.79 below those of raw int16 amplitudes).  The
        default reproduces it (a power of two folded into the filterbank weights: exact); 1.0 gives
        speechpy-on-raw-int16 values.
        overlap_front: run VAD, front end, CMVN and the crop draw of micro-batch k+1 on a second HIP stream
        while the network runs on micro-batch k (device-drawn crops only)."""
        self.eng = get_engine()
        # bench.py sets this to a list: one {kernel name: (start, end) HIP events on the launch stream, "cubes": n} per
        # micro-batch, around every network kernel
        self.kernel_events = None
        self.model = model.to(self.eng.device).eval()
        self.refresh_model()
        self.use_vad, self.vad_threshold, self.normalize = use_vad, int(vad_threshold), bool(normalize)
        self.micro_batch = int(micro_batch)
        # model front end: lmfe(signal, 16000, 0.025, 0.01, 40, 1024)  (load_data.py:64-70, Q14)
        self.spec = spec_from_seconds(c.SAMPLE_RATE, c.FRAME_LEN, c.FRAME_STEP, c.NUM_FFT, c.NUM_COEF, c.NUM_COEF,
                                      _lib.OUT_LMFE, preemph=preemph_cof is not None,
                                      preemph_cof=0.0 if preemph_cof is None else preemph_cof,
                                      input_scale=float(pcm_scale))
        assert crop_rng in ("reference", "device")
        self.crop_rng, self.crop_seed = crop_rng, int(crop_seed)
        self.bad_clips = torch.zeros((1,), dtype=torch.int32, device=self.eng.device)
        # the reference seeds the global NumPy RNG at utils import (utils.py:15) and draws
        # the crop starts from it (utils.py:372); a private RandomState keeps that sequence
        self.rng = np.random.RandomState(crop_seed)
        self.last_stats = {}
        self.overlap_front = bool(overlap_front)
        self._side_stream = None

    def refresh_model(self):
        """Re-snapshot the (BN-folded) inference weights after the model's state changed."""
        self.embedder = self.model.fused_inference()

    def chunks(self, n):
        """(lo, hi) micro-batches of ONE size where possible: the count is ceil(n / micro_batch),
        the size ceil(n / count), and the last one is shifted back to keep that size (it redoes
        a few clips of its neighbour): every launch sequence sees the same shapes."""
        if n <= 0:
            return []
        count = -(-n // self.micro_batch)
        size = -(-n // count)
        spans = [(k * size, min(n, (k + 1) * size)) for k in range(count)]
        lo, hi = spans[-1]
        # (the host RNG of crop_rng="reference" is consumed in clip order: no re-done clips there)
        if hi - lo < size and n >= size and self.crop_rng != "reference":
            spans[-1] = (n - size, n)
        return spans

    # ---- stages ----------------------------------------------------------------------
    def ingest(self, pcm, fs_in, lengths=None):
        """Recordings at another rate / with several channels -> what `embed` takes: int16 mono at
        16 kHz on the device (+ lengths).  pcm: int16 [n, frames] or [n, frames, channels]
        (utils.py:170-173's `librosa.load(..., sr=16000, mono=True)`, batched; ingest.py)."""
        from . import ingest as _ingest
        return _ingest.resample_batch(pcm, fs_in, c.SAMPLE_RATE, lengths=lengths, out_dtype="i16", engine=self.eng)

    def voiced(self, pcm):
        """[n, L] int16 -> (packed voiced samples [n, L] int16, voiced_len [n] i32)."""
        if not self.use_vad:
            n, L = pcm.shape
            return pcm, None
        res = self.eng.vad_energy(pcm, self.vad_threshold, fs=c.SAMPLE_RATE, frame_ms=c.VAD_FRAME_MS,
                                  padding_ms=c.VAD_PADDING_MS, compact=True)
        return res["voiced"], res["voiced_len"]

    def vad(self, pcm, lengths=None, offsets=None, longest=None):
        """The energy VAD WITHOUT the compaction copy: (voiced_len [n] i32, gather) where gather = (src_frame, frame_samples)
        lets the front end read the kept frames where they lie (`features(pcm, voiced_len, gather)`); (None, None) when the
        pipeline runs without VAD.  `voiced()` is the copying form (packed samples, for callers that want them)."""
        if not self.use_vad:
            return (None if lengths is None else lengths), None
        res = self.eng.vad_energy(pcm, self.vad_threshold, fs=c.SAMPLE_RATE, frame_ms=c.VAD_FRAME_MS, padding_ms=c.VAD_PADDING_MS,
                                  lengths=lengths, offsets=offsets, compact="index", longest=longest)
        return res["voiced_len"], (res["src_frame"], res["frame_samples"])

    def features(self, pcm, lengths=None, gather=None):
        feat, n_frames, _ = self.eng.features(pcm, self.spec, lengths=lengths, gather=gather)
        if self.normalize:                                 # utils.CMVN with c.NORMALIZE (utils.py:394-395)
            self.eng.cmvn_(feat, n_frames, variance=True)
        return feat, n_frames

    def draw_crops(self, n_frames_host):
        """idx = randint(T - 80, size=20) per utterance, in order (utils.py:372, Q15)."""
        out = np.empty((len(n_frames_host), c.CUBE_CROPS), dtype=np.int32)
        for i, T in enumerate(n_frames_host):
            if T - c.CUBE_FRAMES <= 0:
                raise ValueError(f"utterance {i} has {T} feature frames; FeatureCube needs more than "
                                 f"{c.CUBE_FRAMES} (numpy randint: low >= high)")
            out[i] = self.rng.randint(int(T) - c.CUBE_FRAMES, size=c.CUBE_CROPS)
        return out

    def cubes(self, feat, crop_idx):
        return self.eng.cube_gather(feat, crop_idx, c.CUBE_FRAMES)

    def crops_and_cubes(self, pcm, first_utt=0, want_cubes=True):
        """The crop starts `embed` would draw for `pcm` ([n, 20] int32 on the host) and, with `want_cubes`, the
        20 x 80 x 40 cubes themselves ([n, 1, 20, 80, 40], device) -- VAD, front end, CMVN, crop draw and gather only, no
        network: what BatchNorm calibration and the parity legs feed to the CPU oracle."""
        pcm = self.eng.to_device(pcm)
        crops, cubes = [], []
        for lo, hi in self.chunks(pcm.shape[0]):
            vlen, gather = self.vad(pcm[lo:hi])
            feat, n_frames = self.features(pcm[lo:hi], vlen, gather)
            if self.crop_rng == "device":
                idx = self.eng.draw_crops(n_frames, c.CUBE_CROPS, c.CUBE_FRAMES, self.crop_seed, first_utt + lo, self.bad_clips)
            else:
                idx = self.draw_crops(n_frames.to("cpu").numpy())
            if want_cubes:
                cubes.append(self.cubes(feat, idx))
            crops.append(idx.cpu().numpy() if hasattr(idx, "cpu") else np.asarray(idx))
        crops = np.concatenate(crops).astype(np.int32) if crops else np.zeros((0, c.CUBE_CROPS), np.int32)
        return (crops, torch.cat(cubes)) if want_cubes else crops

    def embed_cubes(self, cubes):
        """[n, 1, 20, 80, 40] cubes -> [n, 128]: the cube read as feature rows by the first-block kernel (no copy)."""
        return self.embedder(cubes)

    def embed_features(self, feat, crop_idx):
        """features + crop starts -> embeddings.  The cube is never materialised: `svk_c3d2_stage1` reads the
        feature rows and crop starts itself; the other six kernels follow (model.FusedEmbedder.embed_features)."""
        if self.kernel_events is None:
            return self.embedder.embed_features(feat, crop_idx)
        spans = {"cubes": feat.shape[0]}

        def timed(name, fn):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn()
            b.record()
            spans[name] = (a, b)
            return out
        out = self.embedder.embed_features(feat, crop_idx, timed)
        self.kernel_events.append(spans)
        return out

    # ---- whole path ---------------------------------------------------------------------
    def embed(self, pcm, crop_idx=None, return_intermediates=False, first_utt=0):
        """[n, L] int16 PCM (NumPy or CUDA tensor) -> [n, 128] float32 embeddings (device).
        `crop_idx` [n, 20] overrides the RNG draw (parity tests feed both sides the same crops);
        `first_utt` is the global index of row 0 (keys the device-side crop draw)."""
        pcm = self.eng.to_device(pcm)
        emb = torch.empty((pcm.shape[0], 128), dtype=torch.float32, device=self.eng.device)
        inter = []
        spans = self.chunks(pcm.shape[0])
        if (self.overlap_front and crop_idx is None and self.crop_rng == "device" and not return_intermediates
                and len(spans) > 1):
            return self._embed_overlapped(pcm, spans, emb, first_utt)
        for lo, hi in spans:
            chunk = pcm[lo:hi]
            if return_intermediates:                       # the packed voiced samples are part of what is handed back
                voiced, vlen = self.voiced(chunk)
                feat, n_frames = self.features(voiced, vlen)
            else:
                vlen, gather = self.vad(chunk)
                feat, n_frames = self.features(chunk, vlen, gather)
            if crop_idx is None and self.crop_rng == "device":
                idx = self.eng.draw_crops(n_frames, c.CUBE_CROPS, c.CUBE_FRAMES, self.crop_seed,
                                          first_utt + lo, self.bad_clips)
            elif crop_idx is None:
                idx = self.draw_crops(n_frames.to("cpu").numpy())      # tiny D2H: T per utterance
            else:
                idx = np.asarray(crop_idx[lo:hi], dtype=np.int32)
            if return_intermediates:
                cube = self.cubes(feat, idx)
                emb[lo:hi] = self.embed_cubes(cube)
                inter.append({"lo": lo, "hi": hi, "voiced": voiced, "voiced_len": vlen, "feat": feat,
                              "n_frames": n_frames, "crop_idx": idx, "cube": cube})
            else:
                emb[lo:hi] = self.embed_features(feat, idx)
        return (emb, inter) if return_intermediates else emb

    def _embed_overlapped(self, pcm, spans, emb, first_utt):
        """Two HIP streams: the side stream turns micro-batch k+1 into features + crop starts (HBM / VALU work) while the
        main stream runs the MFMA-bound network on micro-batch k."""
        dev = self.eng.device
        main = torch.cuda.current_stream(dev)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=dev)
        side = self._side_stream
        side.wait_stream(main)                     # whatever produced `pcm` is ordered before the side stream

        def stage(k):
            lo, hi = spans[k]
            with torch.cuda.stream(side):
                vlen, gather = self.vad(pcm[lo:hi])
                feat, n_frames = self.features(pcm[lo:hi], vlen, gather)
                idx = self.eng.draw_crops(n_frames, c.CUBE_CROPS, c.CUBE_FRAMES, self.crop_seed, first_utt + lo,
                                          self.bad_clips)
                done = torch.cuda.Event()
                done.record(side)
            return (feat, idx), done

        nxt = stage(0)
        for k, (lo, hi) in enumerate(spans):
            data, done = nxt
            if k + 1 < len(spans):
                nxt = stage(k + 1)
            main.wait_event(done)
            for t in data:
                t.record_stream(main)              # allocated on the side stream, consumed on the main one
            emb[lo:hi] = self.embed_features(*data)
        side.wait_stream(main)
        return emb

    def _ragged_batches(self, lengths, max_batch_samples, max_feature_bytes=1 << 30, max_padding=None):
        """Clip indices sorted by length and cut into batches of at most `micro_batch` clips and `max_batch_samples`
        samples (16-byte-aligned clip slots): a batch costs its own samples, not n x the longest clip -- up to a point: the
        front end enumerates clips x tiles-of-the-LONGEST-clip and the feature buffer is sized the same way, so a batch also
        ends where its padded size (clips x longest) would pass `max_padding` (2 by default) x its real size (the long tail of a VoxCeleb-like
        length distribution otherwise makes one batch of 12 .. 145 s clips that is 88 % padding).  NumPy throughout: the plan
        of 2 048 clips takes ~0.1 ms (a Python loop over clips took 1 ms, with the GPU idle)."""
        if max_padding is None:
            max_padding = float(os.environ.get("SVK_RAGGED_PADDING", "2.0"))
        lengths = np.asarray(lengths, dtype=np.int64)
        n = lengths.size
        if not n:
            return []
        order = np.argsort(lengths, kind="stable")
        slots = (lengths[order] + 7) // 8 * 8
        cum = np.concatenate([[0], np.cumsum(slots)])                   # cum[k] = samples of the first k sorted clips
        out, pos = [], 0
        while pos < n:
            hi = min(n, pos + self.micro_batch)
            # the sample cap: the largest hi with cum[hi] - cum[pos] <= max_batch_samples (at least one clip)
            hi = max(pos + 1, min(hi, int(np.searchsorted(cum, cum[pos] + max_batch_samples, side="right")) - 1))
            # the padding cap: (k - pos) * slots[k - 1] <= max_padding * (cum[k] - cum[pos]) holds at k = pos + 1; take the
            # largest such k up to hi (clips are sorted, so the padded size is clips x the last one)
            k = np.arange(pos + 1, hi + 1)
            ok = (k - pos) * slots[k - 1] <= max_padding * (cum[k] - cum[pos])
            hi = int(k[np.nonzero(ok)[0][-1]])
            out.append((order[pos:hi].tolist(), int(cum[hi] - cum[pos])))
            pos = hi
        # the longest clips would otherwise end up as batches of a handful: a dozen-workgroup front end for a few clips, and every
        # batch has fixed costs (measured, `tools/time_ragged_front.py`: the ONE 145 s clip of the benchmark's 2 048 cost 0.15 ms
        # of VAD + front end + CMVN, as much as 300 clips of 7 s).  A last batch of fewer than 64 clips joins its predecessor
        # (repeatedly) -- when the merged batch stays within `micro_batch` clips, 1.5 x the sample cap, a feature buffer (clips x
        # the LONGEST clip's frames x 40 floats) of `max_feature_bytes` (joining 145 s clips to a full batch of 20 s ones would
        # multiply that buffer), and either the padding cap or a padded size no larger than a full batch's REAL size: tiles past
        # a clip's end leave the front end at once, and a small batch's padding is cheaper than a batch of its own
        while len(out) >= 2 and len(out[-1][0]) < 64:
            merged = len(out[-2][0]) + len(out[-1][0])
            longest = int(lengths[out[-1][0][-1]])
            feat_bytes = merged * max(1, longest // 160) * 40 * 4
            total = out[-2][1] + out[-1][1]
            padded = merged * ((longest + 7) // 8 * 8)
            if (merged <= self.micro_batch and total <= max_batch_samples * 3 // 2 and feat_bytes <= max_feature_bytes
                    and (padded <= max_padding * total or padded <= max_batch_samples)):
                tail = out.pop()
                out[-1] = (out[-1][0] + tail[0], out[-1][1] + tail[1])
            else:
                break
        return out

    @staticmethod
    def _upload_groups(offsets, lengths, n_samples, max_batch_samples):
        """A host arena cut into upload pieces: clips in arena order, a piece closed where the next clip would take it past
        `max_batch_samples` (a single longer clip is a piece of its own).  Returns (groups of clip indices, [lo, hi) sample
        ranges): the ranges tile [0, n_samples) and every clip lies wholly inside its group's range -- also when clips
        OVERLAP in the arena (windows over one recording): a piece ends at the furthest end of its clips, and a clip that
        starts before that point belongs to the same piece."""
        by_pos = np.argsort(offsets, kind="stable")
        starts = offsets[by_pos].astype(np.int64)
        ends = starts + lengths[by_pos]
        groups, pieces, lo, first, reach = [], [], 0, 0, 0
        for q in range(len(by_pos)):
            reach = max(reach, int(ends[q]))
            last = q + 1 == len(by_pos)
            # the next clip may open a new piece only if it starts at or past everything this piece's clips cover
            if last or (int(starts[q + 1]) >= reach and int(ends[q + 1]) - int(starts[first]) > max_batch_samples):
                hi = n_samples if last else int(starts[q + 1])
                groups.append(by_pos[first:q + 1])
                pieces.append((lo, hi))
                lo, first = hi, q + 1
        return groups, pieces

    def _ragged_front(self, dev_buf, offs, lens, longest, rows, spans=None):
        """VAD -> front end -> CMVN statistics -> crop draw of one batch of clips addressed through offsets / lengths (device
        slices) into `dev_buf`; `longest`: the batch's longest clip in samples (host int); rows: the clips' global indices
        (they key the crop draw).  Returns (RAW features [n, T, 40], crop starts [n, 20], CMVN statistics or None): the VAD
        copies nothing (the front end reads the kept frames where they lie, svk_frontend_run's d_src_chunk) and the
        normalisation (utils.py:382-397) is applied by the cube gather to the 20 x 80 rows the network reads, not to every row
        of a clip.  Nothing here touches the host."""
        def timed(name, fn):
            if spans is None:
                return fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn()
            b.record()
            spans.append((name, a, b))
            return out

        dev_lens, gather = lens, None
        if self.use_vad:
            dev_lens, gather = timed("vad", lambda: self.vad(dev_buf, lengths=lens, offsets=offs, longest=longest))
        feat, n_frames, _ = timed("frontend", lambda: self.eng.features(dev_buf, self.spec, lengths=dev_lens, offsets=offs,
                                                                         max_frames=self.spec.num_frames(int(longest)), gather=gather))
        stats = timed("cmvn", lambda: self.eng.cmvn_stats(feat, n_frames, variance=True)) if self.normalize else None
        idx = timed("crops", lambda: self.eng.draw_crops(n_frames, c.CUBE_CROPS, c.CUBE_FRAMES, self.crop_seed, 0, self.bad_clips,
                                                         utt_index=rows))
        return feat, idx, stats

    class _CubeRing:
        """Length-sorted batches are what the front end wants (their feature buffers are sized by the longest clip) and what
        the network does NOT want: the batch of the longest clips is a few dozen cubes, and a persistent kernel that works on
        16-cube groups leaves most of the chip idle there (measured: 0.45 ms of fixed cost per such batch).  So every batch
        leaves only its gathered cubes (the 20 x 80 rows the network will read: 256 KB per clip, utils.py:351-379) in a RING
        of 2 x `step` cubes, and the network runs over `step` cubes as soon as that many have gathered; the first block reads
        the ring as feature rows with crop starts 0, 80, 160 ...  Memory is O(step), whatever the number of clips."""

        def __init__(self, pipe, step, emb, order_dev, spans):
            self.pipe, self.step, self.cap = pipe, int(step), 2 * int(step)
            self.emb, self.order, self.spans = emb, order_dev, spans
            hit = getattr(pipe, "_ring_buf", None)
            if hit is None or hit.shape[0] != self.cap:
                hit = pipe._ring_buf = torch.empty((self.cap, 1, c.CUBE_CROPS, c.CUBE_FRAMES, c.NUM_COEF), dtype=torch.float32,
                                                   device=pipe.eng.device)
            self.cubes = hit
            self.rows = hit.view(self.cap, c.CUBE_CROPS * c.CUBE_FRAMES, c.NUM_COEF)
            self.at = self.done = 0            # cubes gathered / handed to the network so far (absolute counts)

        def _event(self):
            if self.spans is None:
                return None
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev

        def push(self, feat, idx, stats=None):
            n = feat.shape[0]
            assert n <= self.step, "a batch must not exceed the network step"
            a = self._event()
            w = self.at % self.cap
            first = min(n, self.cap - w)
            self.pipe.eng.cube_gather(feat[:first], idx[:first], c.CUBE_FRAMES, out=self.cubes[w:w + first],
                                      stats=None if stats is None else stats[:first])
            if first < n:                        # the batch wraps around the end of the ring
                self.pipe.eng.cube_gather(feat[first:], idx[first:], c.CUBE_FRAMES, out=self.cubes[:n - first],
                                          stats=None if stats is None else stats[first:])
            if a is not None:
                self.spans.append(("gather", a, self._event()))
            self.at += n
            while self.at - self.done >= self.step:
                self._network(self.step)

        def _network(self, n):
            lo = self.done % self.cap              # a multiple of `step`: [lo, lo + n) never wraps
            a = self._event()
            out = self.pipe.embed_features(self.rows[lo:lo + n], self.pipe.embedder.crop_starts(n, self.cubes.device))
            self.emb[self.order[self.done:self.done + n]] = out
            if a is not None:
                self.spans.append(("network", a, self._event()))
            self.done += n

        def finish(self):
            if self.at > self.done:
                self._network(self.at - self.done)

    def embed_ragged(self, clips, max_batch_samples=64 * 1024 * 1024, first_utt=0, spans=None):
        """Clips of DIFFERENT lengths (VoxCeleb1 utterances run from 4 to 145 s): `clips` is a list of 1-D
        int16 arrays on the HOST.  They are sorted by length, packed back to back (16-byte aligned) into batches of
        at most `max_batch_samples` samples and `micro_batch` clips, uploaded, and addressed through the
        offsets / lengths form of the C-ABI.  Embeddings come back in the order of `clips`.  Needs crop_rng='device'.
        `spans`: a list that receives ("vad" | "frontend" | "cmvn" | "crops" | "gather" | "network", start, end) HIP events (bench.py)."""
        if self.crop_rng != "device":
            raise ValueError("embed_ragged needs crop_rng='device'")
        dev = self.eng.device
        emb = torch.empty((len(clips), 128), dtype=torch.float32, device=dev)
        lengths = np.array([len(x) for x in clips], dtype=np.int64)
        batches = self._ragged_batches(lengths, max_batch_samples)
        if not batches:
            return emb
        # Two pinned host buffers + two device buffers: batch k + 1 is packed by 8 host threads (np.copyto releases the
        # GIL) and copied on a side stream while the GPU works on batch k -- the host-side np.zeros + clip-by-clip copy +
        # pageable upload of the first version was 8 x the GPU time of the whole workload.
        cap = max(total for _, total in batches)
        if getattr(self, "_rag_cap", 0) < cap:
            self._rag_cap = cap
            self._rag_pinned = [torch.empty((cap,), dtype=torch.int16).pin_memory() for _ in range(2)]
            self._rag_np = [t.numpy() for t in self._rag_pinned]
            self._rag_dev = [torch.empty((cap,), dtype=torch.int16, device=dev) for _ in range(2)]
            self._rag_stream = torch.cuda.Stream(device=dev)
            from concurrent.futures import ThreadPoolExecutor
            self._rag_threads = int(os.environ.get("SVK_RAGGED_THREADS", "8"))
            self._rag_pool = ThreadPoolExecutor(max_workers=self._rag_threads)
        # every batch's clip order, batch-local offsets and lengths go up ONCE, before the loop: a host array handed to a
        # launch is a synchronous copy that waits for everything queued before it
        order = np.concatenate([np.asarray(b, dtype=np.int64) for b, _ in batches])
        lens_sorted = lengths[order]
        offs_sorted = np.empty_like(order)
        pos = 0
        for batch, _ in batches:
            slots = (lens_sorted[pos:pos + len(batch)] + 7) // 8 * 8
            offs_sorted[pos:pos + len(batch)] = np.concatenate([[0], np.cumsum(slots)[:-1]])
            pos += len(batch)
        order_dev = torch.from_numpy(order).to(dev)
        keys_dev = order_dev + int(first_utt)
        offs_dev = torch.from_numpy(offs_sorted).to(dev)
        lens_dev = torch.from_numpy(lens_sorted.astype(np.int32)).to(dev)
        main = torch.cuda.current_stream(dev)
        copy_stream = self._rag_stream
        copy_stream.synchronize()                          # a previous call's copies may still read the pinned buffers
        copy_stream.wait_stream(main)
        copied = [torch.cuda.Event() for _ in range(2)]
        consumed = [torch.cuda.Event() for _ in range(2)]
        starts = np.concatenate([[0], np.cumsum([len(b) for b, _ in batches])])

        def stage(k):
            batch, total = batches[k]
            slot = k & 1
            if k >= 2:
                consumed[slot].synchronize()               # the GPU is done with what this pinned / device pair held
            offs = offs_sorted[starts[k]:starts[k + 1]]
            dst = self._rag_np[slot]

            def put(lo, hi):
                for q in range(lo, hi):
                    src = np.asarray(clips[batch[q]], dtype=np.int16)
                    np.copyto(dst[offs[q]:offs[q] + src.size], src, casting="no")
            step = -(-len(batch) // self._rag_threads)
            jobs = [self._rag_pool.submit(put, lo, min(len(batch), lo + step)) for lo in range(0, len(batch), step)]
            for jb in jobs:
                jb.result()
            with torch.cuda.stream(copy_stream):
                self._rag_dev[slot][:total].copy_(self._rag_pinned[slot][:total], non_blocking=True)
                copied[slot].record(copy_stream)

        ring = self._CubeRing(self, self.micro_batch, emb, order_dev, spans)
        stage(0)
        for k, (batch, total) in enumerate(batches):
            slot = k & 1
            main.wait_event(copied[slot])
            sl = slice(int(starts[k]), int(starts[k + 1]))
            feat, idx, stats = self._ragged_front(self._rag_dev[slot][:total], offs_dev[sl], lens_dev[sl], int(lens_sorted[sl].max()),
                                                  keys_dev[sl], spans=spans)
            consumed[slot].record(main)
            # the network runs as soon as a full micro-batch of cubes has gathered: its kernels then cover the host-side
            # packing of the next batch
            ring.push(feat, idx, stats)
            if k + 1 < len(batches):
                stage(k + 1)                               # host packing + H2D of the next batch under this batch's kernels
        ring.finish()
        return emb

    def embed_ragged_resident(self, buf, offsets, lengths, max_batch_samples=64 * 1024 * 1024, first_utt=0, spans=None):
        """`embed_ragged` for audio that is ALREADY in one buffer: `buf` is one 1-D int16 array holding every clip, clip k at
        samples [offsets[k], offsets[k] + lengths[k]) (offsets multiples of 8: 16-byte aligned; host arrays; clips may
        overlap, e.g. windows over one recording).
          * a DEVICE tensor: batches are lists of clip indices into that one buffer -- nothing is copied or packed, not even
            the voiced frames (the VAD hands the front end an index of them);
          * a HOST NumPy array (a loader that decodes into one arena): uploaded as it is, no per-clip packing on the host
            (the list form, `embed_ragged`, is bound by that packing: ~20 GB/s of host copy against 54 GB/s of pageable
            upload on the GPU box).  Arenas larger than two batches go up in pieces of ~`max_batch_samples` on a side
            stream from a helper thread, and the clips of piece p run (length-sorted among themselves) while piece p + 1
            travels."""
        if self.crop_rng != "device":
            raise ValueError("embed_ragged_resident needs crop_rng='device'")
        offsets = np.asarray(offsets, dtype=np.int64)
        lengths = np.asarray(lengths, dtype=np.int32)
        if offsets.shape != lengths.shape or (offsets % 8).any():
            raise ValueError("offsets / lengths must have one entry per clip, offsets multiples of 8 samples")
        host = isinstance(buf, np.ndarray)
        if host:
            if buf.ndim != 1 or buf.dtype != np.int16:
                raise ValueError("buf must be a 1-D int16 array")
            buf = np.ascontiguousarray(buf)
            n_samples = buf.size
        else:
            buf = self.eng.to_device(buf)
            if buf.dim() != 1 or buf.dtype != torch.int16:
                raise ValueError("buf must be a 1-D int16 tensor")
            n_samples = buf.numel()
        if len(lengths) and int((offsets + lengths).max()) > n_samples:
            raise ValueError("a clip reaches past the end of buf")
        dev = self.eng.device
        emb = torch.empty((len(lengths), 128), dtype=torch.float32, device=dev)
        if not len(lengths):
            return emb
        # upload groups: clips by arena position, cut where a piece passes max_batch_samples (one group = everything when
        # the buffer is on the device already or small)
        groups, pieces = [np.arange(len(lengths))], [(0, n_samples)]
        if host and n_samples > 2 * max_batch_samples:
            groups, pieces = self._upload_groups(offsets, lengths, n_samples, max_batch_samples)
        # the whole schedule is known from the lengths: plan every group's batches and upload order / offsets / lengths ONCE
        plan = []                                             # (group, clip indices of the batch, longest clip)
        for g, idx in enumerate(groups):
            for batch, _ in self._ragged_batches(lengths[idx], max_batch_samples):
                ids = idx[np.asarray(batch, dtype=np.int64)]
                plan.append((g, ids, int(lengths[ids].max())))
        order = np.concatenate([ids for _, ids, _ in plan]).astype(np.int64)
        order_dev = torch.from_numpy(order).to(dev)
        keys_dev = order_dev + int(first_utt)
        offs_dev = torch.from_numpy(offsets[order]).to(dev)
        lens_dev = torch.from_numpy(lengths[order]).to(dev)
        flags, events, worker = None, None, None
        if host:
            import threading
            if getattr(self, "_up_stream", None) is None:
                self._up_stream = torch.cuda.Stream(device=dev)
            dev_buf = torch.empty((n_samples,), dtype=torch.int16, device=dev)
            flags = [threading.Event() for _ in pieces]
            events = [torch.cuda.Event() for _ in pieces]
            self._up_stream.wait_stream(torch.cuda.current_stream(dev))
            src = torch.from_numpy(buf)

            failure = []

            def upload():   # (pageable copies hold their calling thread: a helper thread, so that the main one keeps launching)
                try:
                    with torch.cuda.stream(self._up_stream):
                        for g, (a, b) in enumerate(pieces):
                            dev_buf[a:b].copy_(src[a:b], non_blocking=True)
                            events[g].record(self._up_stream)
                            flags[g].set()
                except BaseException as err:   # surfaces in the caller's thread; nobody is left waiting
                    failure.append(err)
                    for f in flags:
                        f.set()
            worker = threading.Thread(target=upload, daemon=True)
            worker.start()
            buf = dev_buf
        # with pieces still travelling the network runs per micro-batch of gathered cubes (it covers the next piece's upload);
        # otherwise over micro-batches as large as the main path's
        step = self.micro_batch if len(groups) > 1 else max(self.micro_batch, 4096)
        ring = self._CubeRing(self, step, emb, order_dev, spans)
        main = torch.cuda.current_stream(dev)
        pos, seen = 0, -1
        for g, ids, longest in plan:
            if flags is not None and g != seen:
                flags[g].wait()
                if failure:
                    worker.join()
                    raise failure[0]
                main.wait_event(events[g])
                seen = g
            sl = slice(pos, pos + len(ids))
            feat, idx, stats = self._ragged_front(buf, offs_dev[sl], lens_dev[sl], longest, keys_dev[sl], spans=spans)
            ring.push(feat, idx, stats)
            pos += len(ids)
        if worker is not None:
            worker.join()
        ring.finish()
        return emb

    def embed_host(self, pcm_host, first_utt=0):
        """Host-fed variant of `embed`: `pcm_host` is a [n, L] int16 NumPy array (e.g. decoded WAVs) or a CPU torch
        tensor, pinned or not.  Micro-batches go through a copy stream and two device buffers, so the H2D copy of batch
        k + 1 overlaps the kernels of batch k (96 kB per 3 s clip over PCIe).  A HELPER THREAD issues the
        copies: a pageable upload holds its calling thread (the runtime stages it through its own pinned chunks, 54 GB/s on
        the GPU box -- faster than the 20 GB/s at which this process could copy into a pinned buffer of its own, the
        round-2 form), and the main thread keeps launching kernels."""
        if self.crop_rng != "device":
            raise ValueError("embed_host overlaps copies with compute and needs crop_rng='device'")
        import threading
        src = pcm_host if isinstance(pcm_host, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pcm_host))
        if src.dtype != torch.int16 or src.dim() != 2 or src.is_cuda:
            raise ValueError("pcm_host must be a [n, L] int16 array in host memory")
        n, L = src.shape
        dev = self.eng.device
        spans = self.chunks(n)
        emb = torch.empty((n, 128), dtype=torch.float32, device=dev)
        if not spans:
            return emb
        size = max(hi - lo for lo, hi in spans)
        key = (size, L)
        if getattr(self, "_host_key", None) != key:
            self._host_key = key
            self._staged = [torch.empty((size, L), dtype=torch.int16, device=dev) for _ in range(2)]
            self._copy_stream = torch.cuda.Stream(device=dev)
        staged, copy_stream = self._staged, self._copy_stream
        main = torch.cuda.current_stream(dev)
        # The staged buffers outlive the call: kernels of a previous call on `main` may still READ them.
        copy_stream.wait_stream(main)
        copied = [torch.cuda.Event() for _ in spans]
        consumed = [torch.cuda.Event() for _ in spans]
        issued = [threading.Event() for _ in spans]      # host side: copy k is queued / batch k's kernels are queued
        launched = [threading.Event() for _ in spans]
        failure = []

        def uploader():
            try:
                with torch.cuda.stream(copy_stream):
                    for k, (lo, hi) in enumerate(spans):
                        if k >= 2:                        # the slot's previous batch must have been consumed by the GPU
                            launched[k - 2].wait()
                            copy_stream.wait_event(consumed[k - 2])
                        staged[k & 1][:hi - lo].copy_(src[lo:hi], non_blocking=True)
                        copied[k].record(copy_stream)
                        issued[k].set()
            except BaseException as err:                  # surface it in the caller's thread
                failure.append(err)
                for ev in issued:
                    ev.set()

        worker = threading.Thread(target=uploader, daemon=True)
        worker.start()
        try:
            for k, (lo, hi) in enumerate(spans):
                issued[k].wait()
                if failure:
                    raise failure[0]
                main.wait_event(copied[k])
                chunk = staged[k & 1][:hi - lo]
                vlen, gather = self.vad(chunk)
                feat, n_frames = self.features(chunk, vlen, gather)
                idx = self.eng.draw_crops(n_frames, c.CUBE_CROPS, c.CUBE_FRAMES, self.crop_seed, first_utt + lo,
                                          self.bad_clips)
                emb[lo:hi] = self.embed_features(feat, idx)
                consumed[k].record(main)
                launched[k].set()
        finally:
            for ev in launched:                           # never leave the helper waiting
                ev.set()
            worker.join()
        return emb

    def score(self, test_emb, enroll_emb):
        return self.eng.cosine_scores(test_emb, enroll_emb)


def enroll_last_utterance(embeddings, speaker_ids):
    """Speaker model = embedding of that speaker's LAST listed utterance: the reference
    overwrites `{id}.pt` on every utterance, no averaging (Q17, model.py:374-388).
    Returns (sorted unique ids, row index of each speaker's model)."""
    speaker_ids = np.asarray(speaker_ids)
    uniq = np.unique(speaker_ids)
    last = np.array([np.nonzero(speaker_ids == s)[0][-1] for s in uniq], dtype=np.int64)
    return uniq, last
