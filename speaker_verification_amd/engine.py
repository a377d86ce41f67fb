"""Device engine: the thin host layer between Python callers and libsvk.so.

One `Engine` per process and GPU.  torch supplies device memory and streams
(plumbing); every computation below is a hand-written gfx950 kernel reached
through the C-ABI.  Methods take torch CUDA tensors (zero-copy) or NumPy arrays
(uploaded) and return torch CUDA tensors; the drop-in modules
(`speechpy/`, `vad.py`, `evaluation.py`, `siamese.py`) convert to the
reference's NumPy return types.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import FrontendCfg, check


def _torch():
    import torch
    return torch


class FrontendSpec:
    """Hashable description of one front-end configuration (SURVEY.md section 5:
    'a small frozen FrontEndConfig')."""

    __slots__ = ("fs", "frame_len", "frame_stride", "nfft", "num_filters", "num_ceps", "out_kind",
                 "dc_elimination", "low_freq", "high_freq", "preemph", "preemph_shift", "preemph_cof",
                 "input_scale")

    def __init__(self, fs, frame_len, frame_stride, nfft, num_filters, num_ceps, out_kind,
                 dc_elimination=True, low_freq=0, high_freq=None, preemph=False, preemph_shift=1,
                 preemph_cof=0.98, input_scale=1.0):
        """input_scale: amplitude factor applied to the PCM before anything else (1/32768 reads int16
        PCM as the float signal librosa.load hands the reference's lmfe call, load_data.py:50-70)."""
        self.input_scale = float(input_scale)
        self.fs, self.frame_len, self.frame_stride, self.nfft = fs, int(frame_len), int(frame_stride), int(nfft)
        self.num_filters, self.num_ceps, self.out_kind = int(num_filters), int(num_ceps), int(out_kind)
        self.dc_elimination, self.low_freq, self.high_freq = bool(dc_elimination), low_freq, high_freq
        self.preemph, self.preemph_shift, self.preemph_cof = bool(preemph), int(preemph_shift), float(preemph_cof)

    def key(self):
        return tuple(getattr(self, s) for s in self.__slots__)

    @property
    def num_cols(self):
        return self.num_ceps if self.out_kind == _lib.OUT_MFCC else self.num_filters

    def num_frames(self, n_samples):
        """floor((L - flen) / stride), never negative (processing.py:115-116, Q3)."""
        return max(0, int(math.floor((n_samples - self.frame_len) / float(self.frame_stride)))) \
            if n_samples >= self.frame_len else 0

    def c_struct(self):
        return FrontendCfg(self.frame_len, self.frame_stride, self.nfft, self.num_filters,
                           max(1, self.num_ceps), self.out_kind, int(self.dc_elimination), int(self.preemph),
                           self.preemph_shift, self.preemph_cof, self.input_scale)


def spec_from_seconds(fs, frame_length, frame_stride, nfft, num_filters, num_ceps, out_kind, **kw):
    """frame sizes as the reference derives them (processing.py:94-98)."""
    flen = int(np.round(fs * frame_length))
    stride = int(float(np.round(fs * frame_stride)))
    return FrontendSpec(fs, flen, stride, nfft, num_filters, num_ceps, out_kind, **kw)


class Engine:
    def __init__(self, device=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("speaker_verification_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = _lib.load()
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        handle = C.c_void_p()
        check(self.lib.svk_create(self.device_index, C.byref(handle)))
        self.ctx = handle
        self._plans = {}
        info = (C.c_int64 * 4)()
        check(self.lib.svk_device_info(self.ctx, info), self.ctx)
        self.num_cu, self.clock_khz, self.lds_per_cu, self.wave = (int(v) for v in info)

    def __del__(self):
        try:
            for hit in self._plans.values():
                if not isinstance(hit, str):
                    self.lib.svk_frontend_plan_destroy(hit[0])
            if self.ctx:
                self.lib.svk_destroy(self.ctx)
        except Exception:
            pass

    # ---- plumbing -----------------------------------------------------------
    def _stream(self):
        torch = _torch()
        stream = torch.cuda.current_stream(self.device)
        check(self.lib.svk_set_stream(self.ctx, C.c_void_p(stream.cuda_stream)), self.ctx)

    def to_device(self, x, dtype=None):
        """torch CUDA tensor (contiguous) from a NumPy array / tensor."""
        torch = _torch()
        if isinstance(x, torch.Tensor):
            t = x.to(self.device)
        else:
            x = np.ascontiguousarray(x)
            if not x.flags.writeable:            # e.g. np.frombuffer views: torch wants a writable array
                x = x.copy()
            t = torch.from_numpy(x).to(self.device)
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous()

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def synchronize(self):
        check(self.lib.svk_sync(self.ctx), self.ctx)

    # ---- fused front end -------------------------------------------------------
    def plan(self, spec):
        from .speechpy import feature as _feature
        key = spec.key()
        hit = self._plans.get(key)
        if isinstance(hit, str):                 # a configuration the fused kernel refused before: a FRESH exception
            raise _lib.SvkError(_lib.SVK_ERR_UNSUPPORTED, hit)   # each time (a cached one would chain every caller's
        if hit is None:                                          # traceback -- and its frames' inputs -- forever)
            bank = np.ascontiguousarray(
                _feature.filterbanks(spec.num_filters, spec.nfft // 2 + 1, spec.fs, spec.low_freq,
                                     spec.high_freq or spec.fs / 2), dtype=np.float64)
            cfg = spec.c_struct()
            handle = C.c_void_p()
            try:
                check(self.lib.svk_frontend_plan_create(self.ctx, C.byref(cfg),
                                                        bank.ctypes.data_as(C.POINTER(C.c_double)),
                                                        C.byref(handle)), self.ctx)
            except _lib.SvkError as err:
                if err.code == _lib.SVK_ERR_UNSUPPORTED:
                    self._plans[key] = err.message or "unsupported front-end configuration"
                raise
            hit = (handle, cfg)
            self._plans[key] = hit
        return hit

    def features(self, pcm, spec, lengths=None, offsets=None, clip_len=None, max_frames=None,
                 want_energy=False, gather=None):
        """Batched front end.

        pcm     : [n_utt, L] (uniform clips) or, with `offsets`, a 1-D concatenation;
                  int16 or float32, NumPy or CUDA tensor
        lengths : optional [n_utt] int32 samples per clip (device or host)
        gather  : (src_frame [n_utt, F] i32, frame_samples) of vad_energy(..., compact="index"): the clips' VOICED frames
                  are read where they lie in `pcm` (lengths = voiced_len), nothing was copied (int16 PCM, fused kernel only)
        returns (feat [n_utt, max_frames, cols] f32, n_frames [n_utt] i32, energy or None)
        """
        torch = _torch()
        try:
            handle, _ = self.plan(spec)
        except _lib.SvkError as err:
            if err.code != _lib.SVK_ERR_UNSUPPORTED or gather is not None:
                raise
            # what the fused kernel does not cover (other fft lengths, > 64 filters, a bank past bin
            # nfft/4 of 1024 or nfft/2 of 512, frames too long for a CU's LDS): the same stages as
            # separate kernels, clip by clip
            return self._features_staged(pcm, spec, lengths, offsets, clip_len, max_frames, want_energy)
        pcm = self.to_device(pcm)
        if pcm.dtype == torch.int16:
            kind = _lib.PCM_I16
        else:
            pcm = pcm.to(torch.float32) if pcm.dtype != torch.float32 else pcm
            kind = _lib.PCM_F32
        if lengths is not None:
            lengths = self.to_device(lengths, torch.int32)
        if offsets is not None:
            if lengths is None:
                raise ValueError("offsets need lengths")
            offsets = self.to_device(offsets, torch.int64)
            n_utt, stride, length = offsets.numel(), 0, 0
            # (a device-side max would be a host round trip per batch: only when the caller did not size the output)
            longest = (int(lengths.max().item()) if n_utt else 0) if max_frames is None else 0
        else:
            if pcm.dim() == 1:
                pcm = pcm[None]
            n_utt, stride = pcm.shape[0], pcm.shape[1]
            length = stride if clip_len is None else int(clip_len)
            longest = length
        if max_frames is None:
            max_frames = spec.num_frames(longest)
        cols = spec.num_cols
        feat = torch.empty((n_utt, max_frames, cols), dtype=torch.float32, device=self.device)
        n_frames = torch.empty((n_utt,), dtype=torch.int32, device=self.device)
        energy = torch.empty((n_utt, max_frames), dtype=torch.float32, device=self.device) if want_energy else None
        src, chunk, cstride = None, 0, 0
        if gather is not None:
            src, chunk = gather
            if lengths is None or src.dtype != torch.int32 or src.dim() != 2 or src.shape[0] != n_utt or not src.is_contiguous():
                raise ValueError("gather wants (src_frame [n_utt, F] int32, frame_samples) and the voiced lengths")
            cstride = int(src.shape[1])
        self._stream()
        check(self.lib.svk_frontend_run(self.ctx, handle, self._ptr(pcm), kind, self._ptr(offsets),
                                        self._ptr(lengths), stride, length, n_utt, max_frames,
                                        self._ptr(feat), self._ptr(energy), self._ptr(n_frames),
                                        self._ptr(src), int(chunk), cstride), self.ctx)
        return feat, n_frames, energy

    def _features_staged(self, pcm, spec, lengths, offsets, clip_len, max_frames, want_energy):
        """`features` for any configuration: svk_preemphasis -> svk_stack_frames -> svk_spectrum ->
        svk_mel_features per clip (feature.py:156-219 stage by stage).  Same outputs and layout."""
        torch = _torch()
        from .speechpy import feature as _feature
        pcm = self.to_device(pcm)
        if pcm.dtype not in (torch.int16, torch.float32):
            pcm = pcm.to(torch.float32)
        if offsets is not None:
            if lengths is None:
                raise ValueError("offsets need lengths")
            offs = [int(v) for v in torch.as_tensor(offsets).cpu().tolist()]
            lens = [int(v) for v in torch.as_tensor(lengths).cpu().tolist()]
            clips = [pcm.reshape(-1)[o:o + n] for o, n in zip(offs, lens)]
        else:
            if pcm.dim() == 1:
                pcm = pcm[None]
            full = pcm.shape[1] if clip_len is None else int(clip_len)
            lens = [full] * pcm.shape[0] if lengths is None else \
                [int(v) for v in torch.as_tensor(lengths).cpu().tolist()]
            clips = [pcm[i, :n] for i, n in enumerate(lens)]
        if max_frames is None:
            max_frames = spec.num_frames(max(lens) if lens else 0)
        bank = _feature.filterbanks(spec.num_filters, spec.nfft // 2 + 1, spec.fs, spec.low_freq,
                                    spec.high_freq or spec.fs / 2)
        bank_dev = self.to_device(bank, torch.float32)
        cols = spec.num_cols
        feat = torch.zeros((len(clips), max_frames, cols), dtype=torch.float32, device=self.device)
        n_frames = torch.zeros((len(clips),), dtype=torch.int32)
        energy = torch.zeros((len(clips), max_frames), dtype=torch.float32, device=self.device) if want_energy else None
        for i, clip in enumerate(clips):
            T = min(spec.num_frames(int(clip.numel())), max_frames)
            n_frames[i] = T
            if T <= 0:
                continue
            sig = self.preemphasis(clip, spec.preemph_shift, spec.preemph_cof) if spec.preemph else clip.to(torch.float32)
            if spec.input_scale != 1.0:
                sig = sig * spec.input_scale
            frames = self.stack_frames(sig, spec.frame_len, spec.frame_stride, T)
            power = self.spectrum(frames, spec.nfft, power=True)
            f, e = self.mel_features(power, bank_dev, spec.out_kind, spec.num_ceps, spec.dc_elimination, want_energy)
            feat[i, :T] = f
            if want_energy:
                energy[i, :T] = e
        return feat, n_frames.to(self.device), energy

    # ---- stage-level kernels ------------------------------------------------------
    def preemphasis(self, signal, shift=1, cof=0.98):
        torch = _torch()
        x = self.to_device(signal)
        if x.dtype == torch.int16:
            kind = _lib.PCM_I16
        else:
            x = x.to(torch.float32)
            kind = _lib.PCM_F32
        out = torch.empty(x.shape, dtype=torch.float32, device=self.device)
        self._stream()
        check(self.lib.svk_preemphasis(self.ctx, self._ptr(x), kind, x.numel(), int(shift), float(cof),
                                       self._ptr(out)), self.ctx)
        return out

    def stack_frames(self, sig, frame_len, stride, n_frames, window=None):
        torch = _torch()
        x = self.to_device(sig, torch.float32)
        win = self.to_device(window, torch.float32) if window is not None else None
        out = torch.empty((max(n_frames, 0), frame_len), dtype=torch.float32, device=self.device)
        self._stream()
        check(self.lib.svk_stack_frames(self.ctx, self._ptr(x), x.numel(), int(frame_len), int(stride),
                                        int(max(n_frames, 0)), self._ptr(win), self._ptr(out)), self.ctx)
        return out

    def spectrum(self, frames, nfft, power):
        torch = _torch()
        fr = self.to_device(frames, torch.float32)
        if fr.dim() != 2:
            raise ValueError("frames must be (num_frames, frame_len)")
        out = torch.empty((fr.shape[0], nfft // 2 + 1), dtype=torch.float32, device=self.device)
        self._stream()
        check(self.lib.svk_spectrum(self.ctx, self._ptr(fr), fr.shape[0], fr.shape[1], int(nfft), int(bool(power)),
                                    self._ptr(out)), self.ctx)
        return out

    def cmvn_(self, feat, n_frames=None, variance=False):
        """In place on feat [n_utt, max_frames, cols] (or [rows, cols] = one clip)."""
        torch = _torch()
        if not (isinstance(feat, torch.Tensor) and feat.is_cuda and feat.dtype == torch.float32
                and feat.is_contiguous()):
            raise ValueError("cmvn_ works in place on a contiguous float32 CUDA tensor")
        shape = feat.shape if feat.dim() == 3 else (1,) + tuple(feat.shape)
        nf = self.to_device(n_frames, torch.int32) if n_frames is not None else None
        self._stream()
        check(self.lib.svk_cmvn(self.ctx, self._ptr(feat), shape[0], shape[1], shape[2], self._ptr(nf),
                                int(bool(variance))), self.ctx)
        return feat

    def cmvn_stats(self, feat, n_frames=None, variance=False):
        """svk_cmvn_stats: per-clip mean and 1 / (std + 2^-30) of feat [n_utt, max_frames, cols] as float64 [n_utt, 2, cols],
        without touching feat (cube_gather(..., stats=...) applies them to the rows it copies)."""
        torch = _torch()
        if not (isinstance(feat, torch.Tensor) and feat.is_cuda and feat.dtype == torch.float32 and feat.is_contiguous()
                and feat.dim() == 3):
            raise ValueError("cmvn_stats wants a contiguous float32 CUDA tensor [n_utt, max_frames, cols]")
        nf = self.to_device(n_frames, torch.int32) if n_frames is not None else None
        stats = torch.zeros((feat.shape[0], 2, feat.shape[2]), dtype=torch.float64, device=self.device)
        self._stream()
        check(self.lib.svk_cmvn_stats(self.ctx, self._ptr(feat), feat.shape[0], feat.shape[1], feat.shape[2], self._ptr(nf),
                                      int(bool(variance)), self._ptr(stats)), self.ctx)
        return stats

    def mel_features(self, power, bank, out_kind, num_ceps=13, dc_elimination=True, want_energy=False):
        """General mel / log / DCT stage on a device power spectrum [T, bins] (any fft length)."""
        torch = _torch()
        p = self.to_device(power, torch.float32)
        b = self.to_device(bank, torch.float32)
        T, bins = p.shape
        nf = b.shape[0]
        cols = num_ceps if out_kind == _lib.OUT_MFCC else nf
        feat = torch.empty((T, cols), dtype=torch.float32, device=self.device)
        energy = torch.empty((T,), dtype=torch.float32, device=self.device) if want_energy else None
        self._stream()
        check(self.lib.svk_mel_features(self.ctx, self._ptr(p), T, bins, self._ptr(b), nf, int(out_kind),
                                        int(num_ceps), int(bool(dc_elimination)), self._ptr(feat),
                                        self._ptr(energy)), self.ctx)
        return feat, energy

    def cmvnw(self, feat, win_size=301, variance=False, n_frames=None):
        """Sliding-window CMVN of feat [n_utt, max_frames, cols] (or [rows, cols]); returns a new tensor."""
        torch = _torch()
        x = self.to_device(feat, torch.float32)
        shape = x.shape if x.dim() == 3 else (1,) + tuple(x.shape)
        nf = self.to_device(n_frames, torch.int32) if n_frames is not None else None
        out = x.clone()                         # rows past n_frames keep the input
        tmp = torch.empty_like(x) if variance else None
        self._stream()
        check(self.lib.svk_cmvnw(self.ctx, self._ptr(x), shape[0], shape[1], shape[2], self._ptr(nf), int(win_size),
                                 int(bool(variance)), self._ptr(tmp), self._ptr(out)), self.ctx)
        return out

    def derivative(self, feat, delta):
        torch = _torch()
        x = self.to_device(feat, torch.float32)
        out = torch.empty_like(x)
        self._stream()
        check(self.lib.svk_derivative(self.ctx, self._ptr(x), x.numel() // x.shape[-1], x.shape[-1], int(delta),
                                      self._ptr(out)), self.ctx)
        return out

    def log_power_(self, power, normalize=True):
        """In place: 10 log10(max(p, 1e-20)) [- global max]."""
        self._stream()
        check(self.lib.svk_log_power(self.ctx, self._ptr(power), power.numel(), int(bool(normalize))), self.ctx)
        return power

    def vad_energy(self, pcm, threshold, fs=16000, frame_ms=30, padding_ms=300, lengths=None, compact=True,
                   want_segments=False, frame_samples=None, ring_len=None, offsets=None, voiced_out=None, longest=None):
        """pcm [n_utt, L] int16 (or, with `offsets` + `lengths`, a 1-D concatenation of ragged clips) ->
        dict(keep [n, F] u8, n_vad_frames [n] i32, voiced (same layout as pcm; a clip's samples past its
        voiced_len are unspecified) i16, voiced_len [n] i32, seg [n, F] i32, src_frame).  compact=True copies the kept
        frames to the front of `voiced`; compact="index" copies nothing and returns src_frame [n, F] i32 instead (entry q =
        the q-th kept frame: `features(pcm, ..., lengths=voiced_len, gather=(src_frame, frame_samples))` reads through it).  `longest`: the longest clip in samples when
        the caller knows it (device-side lengths would otherwise cost a host round trip to size the outputs)."""
        torch = _torch()
        x = self.to_device(pcm)
        if x.dtype != torch.int16:
            raise TypeError("VAD works on int16 PCM (vad.py:16-17 asserts 16-bit mono)")
        fsamp = int(frame_samples) if frame_samples else int(fs * (frame_ms / 1000.0) * 2) // 2   # vad.py:50
        ring_len = int(ring_len) if ring_len else int(padding_ms / frame_ms)                       # vad.py:81
        ring_thresh = int(math.floor(0.9 * ring_len))                    # count > 0.9 * maxlen, vad.py:99,117
        lens = self.to_device(lengths, torch.int32) if lengths is not None else None
        if offsets is not None:
            if lens is None:
                raise ValueError("offsets need lengths")
            offs = self.to_device(offsets, torch.int64)
            n_utt, stride = offs.numel(), 0
            if longest is not None:
                longest = int(longest)
            elif not n_utt:
                longest = 0
            elif isinstance(lengths, np.ndarray):          # host lengths: no device round trip
                longest = int(lengths.max())
            else:
                longest = int(lens.max().item())
        else:
            offs = None
            if x.dim() == 1:
                x = x[None]
            n_utt, stride = x.shape
            longest = stride
        max_vf = max(1, (2 * longest - 1) // (2 * fsamp)) if longest > 0 else 1
        keep = torch.empty((n_utt, max_vf), dtype=torch.uint8, device=self.device)
        nvf = torch.empty((n_utt,), dtype=torch.int32, device=self.device)
        seg = torch.empty((n_utt, max_vf), dtype=torch.int32, device=self.device) if want_segments else None
        # only [:voiced_len] of a clip is defined (no 98 MB memset); `voiced_out`: a caller-owned buffer of pcm's shape,
        # reused across batches that address ONE resident concatenation through offsets
        index = compact == "index"
        if index:
            voiced = None
        elif compact and voiced_out is not None:
            if voiced_out.shape != x.shape or voiced_out.dtype != torch.int16 or not voiced_out.is_contiguous():
                raise ValueError("voiced_out must be a contiguous int16 tensor of pcm's shape")
            voiced = voiced_out
        else:
            voiced = torch.empty_like(x) if compact else None
        vlen = torch.empty((n_utt,), dtype=torch.int32, device=self.device) if compact else None
        src = torch.empty((n_utt, max_vf), dtype=torch.int32, device=self.device) if index else None
        self._stream()
        check(self.lib.svk_vad_energy(self.ctx, self._ptr(x), self._ptr(offs), self._ptr(lens), stride, longest, n_utt,
                                      fsamp, ring_len, ring_thresh, int(threshold), max_vf, self._ptr(keep),
                                      self._ptr(seg), self._ptr(nvf), self._ptr(voiced), self._ptr(vlen), self._ptr(src)), self.ctx)
        return {"keep": keep, "n_vad_frames": nvf, "voiced": voiced, "voiced_len": vlen, "seg": seg,
                "frame_samples": fsamp, "src_frame": src}

    def draw_crops(self, n_frames, n_crops=20, crop_frames=80, seed=12345, first_utt=0, bad_count=None,
                   utt_index=None):
        """[n] i32 frame counts (device) -> [n, n_crops] i32 crop starts drawn on the device, keyed by
        the clip's global index: `utt_index[u]` if given, else first_utt + u."""
        torch = _torch()
        nf = self.to_device(n_frames, torch.int32)
        gi = self.to_device(utt_index, torch.int64) if utt_index is not None else None
        idx = torch.empty((nf.numel(), n_crops), dtype=torch.int32, device=self.device)
        self._stream()
        check(self.lib.svk_cube_draw_crops(self.ctx, self._ptr(nf), nf.numel(), int(first_utt), self._ptr(gi), n_crops,
                                           crop_frames, int(seed) & 0xFFFFFFFFFFFFFFFF, self._ptr(idx),
                                           self._ptr(bad_count)), self.ctx)
        return idx

    def cube_gather(self, feat, crop_idx, crop_frames=80, out=None, stats=None):
        """feat [n, T, C] + crop_idx [n, n_crops] -> [n, 1, n_crops, crop_frames, C] (utils.py:364-379).  `stats` (cmvn_stats):
        the copied rows are CMVN-normalised on the way (svk_cube_gather_cmvn) -- feat itself stays raw."""
        torch = _torch()
        feat = self.to_device(feat, torch.float32)
        idx = self.to_device(crop_idx, torch.int32)
        n, T, Cc = feat.shape
        n_crops = idx.shape[1]
        if out is None:
            out = torch.empty((n, 1, n_crops, crop_frames, Cc), dtype=torch.float32, device=self.device)
        self._stream()
        if stats is not None:
            if stats.dtype != torch.float64 or tuple(stats.shape) != (n, 2, Cc) or not stats.is_contiguous():
                raise ValueError("stats must be the float64 [n, 2, cols] tensor of cmvn_stats")
            check(self.lib.svk_cube_gather_cmvn(self.ctx, self._ptr(feat), n, T, Cc, self._ptr(idx), n_crops, crop_frames,
                                                self._ptr(stats), self._ptr(out)), self.ctx)
            return out
        check(self.lib.svk_cube_gather(self.ctx, self._ptr(feat), n, T, Cc, self._ptr(idx), n_crops, crop_frames,
                                       self._ptr(out)), self.ctx)
        return out

    def c3d2_stage1(self, feat, crop_idx, tables, crop_frames=80):
        """svk_c3d2_stage1: feature rows + crop starts -> the activation after C3D2's first block (conv1_1, conv1_2,
        pool1 with their BN + PReLU): [n, 16, 36, 18, 16] f32, channels last.  Two-piece f16 products on
        v_mfma_f32_16x16x32_f16 (x = h + l, three piece products per f32 product, f32 accumulation: ~1e-6 of the scale from the
        f32 form); tables: `FusedEmbedder.stage1_tables()`."""
        torch = _torch()
        feat = self.to_device(feat, torch.float32)
        idx = self.to_device(crop_idx, torch.int32)
        n, T, Cc = feat.shape
        w1frag, bias1, slope1, w2frag, bias2, slope2 = tables[:6]
        if w1frag.dtype != torch.float16 or w2frag.dtype != torch.float16 or tuple(w1frag.shape) != (2, 64, 8) or tuple(w2frag.shape) != (14, 2, 64, 8):
            raise ValueError("c3d2_stage1 wants the half-pair weight blocks of FusedEmbedder.stage1_tables()")
        slope01 = bool(tables[6]) if len(tables) > 6 else False          # every slope in [0, 1]: the two-instruction PReLU
        out = torch.empty((n, 16, 36, 18, 16), dtype=torch.float32, device=self.device)
        self._stream()
        check(self.lib.svk_c3d2_stage1(self.ctx, self._ptr(feat), n, T, Cc, self._ptr(idx), idx.shape[1], crop_frames,
                                       self._ptr(w1frag), self._ptr(bias1), self._ptr(slope1), self._ptr(w2frag), self._ptr(bias2),
                                       self._ptr(slope2), 2 if slope01 else 0, self._ptr(out)), self.ctx)
        return out

    def c3d2_stage2(self, act1, tables):
        """svk_c3d2_stage2: [n, 16, 36, 18, 16] (svk_c3d2_stage1's output) -> conv2_1 -> conv2_2 -> pool2 with their
        BN + PReLU -> [n, 12, 15, 7, 32] f32 (channels last), both convolutions through two-piece f16 products."""
        torch = _torch()
        n = act1.shape[0]
        if tuple(act1.shape[1:]) != (16, 36, 18, 16) or not act1.is_contiguous():
            raise ValueError("c3d2_stage2 wants the activation [n, 16, 36, 18, 16]")
        w21, b21, s21, w22, b22, s22 = tables[:6]
        if w21.dtype != torch.float16 or w22.dtype != torch.float16 or tuple(w21.shape) != (2, 6, 2, 64, 8) or tuple(w22.shape) != (2, 24, 2, 64, 8):
            raise ValueError("c3d2_stage2 wants the half-pair weight blocks of FusedEmbedder.stage2_tables()")
        slope01 = bool(tables[6]) if len(tables) > 6 else False
        act2 = torch.empty((n, 14, 36, 14, 32), dtype=torch.float32, device=self.device)   # scratch: the 14 columns of conv2_1 that pool2 leaves alive
        out = torch.empty((n, 12, 15, 7, 32), dtype=torch.float32, device=self.device)
        self._stream()
        check(self.lib.svk_c3d2_stage2(self.ctx, self._ptr(act1), n, self._ptr(w21), self._ptr(b21), self._ptr(s21),
                                       self._ptr(w22), self._ptr(b22), self._ptr(s22),
                                       2 if slope01 else 0,
                                       self._ptr(act2), self._ptr(out)), self.ctx)
        return out

    def c3d2_conv31(self, act, tables):
        """svk_c3d2_conv31: [n, 12, 15, 7, 32] (svk_c3d2_stage2's output) -> conv3_1 + BN + PReLU -> chunked, column-major
        [n, 10 d, 8 chunks, 5 w, 15 h, 8] f32: what svk_c3d2_conv32t stages from."""
        torch = _torch()
        n = act.shape[0]
        if tuple(act.shape[1:]) != (12, 15, 7, 32) or not act.is_contiguous():
            raise ValueError("c3d2_conv31 wants the activation [n, 12, 15, 7, 32]")
        wfrag, bias, slope = tables[:3]
        slope01 = bool(tables[3]) if len(tables) > 3 else False
        out = torch.empty((n, 10, 8, 5, 15, 8), dtype=torch.float32, device=self.device)
        self._stream()
        check(self.lib.svk_c3d2_conv31(self.ctx, self._ptr(act), n, self._ptr(wfrag), self._ptr(bias), self._ptr(slope),
                                       2 if slope01 else 0, self._ptr(out)), self.ctx)
        return out

    def c3d2_conv32t(self, act, tables):
        """svk_c3d2_conv32t: chunked, column-major [n, 10, 8, 5, 15, 8] (svk_c3d2_conv31's output) -> conv3_2 + BN + PReLU ->
        chunked [n, 8 d, 8 chunks, 45 = 9 h x 5 w, 8] (the shape of the last block: M tile = one position of 16 cubes)."""
        if tuple(act.shape[1:]) != (10, 8, 5, 15, 8) or not act.is_contiguous():
            raise ValueError("c3d2_conv32t wants the chunked activation [n, 10, 8, 5, 15, 8]")
        return self._c3d2_tail_conv(self.lib.svk_c3d2_conv32t, act, tables, (8, 8, 45, 8), (4, 2, 21, 2, 64, 8), "float16")

    def _c3d2_tail_conv(self, fn, act, tables, out_shape, w_shape, w_dtype):
        torch = _torch()
        n = act.shape[0]
        wfrag, bias, slope = tables[:3]
        if tuple(wfrag.shape) != w_shape or wfrag.dtype != getattr(torch, w_dtype) or not wfrag.is_contiguous():
            raise ValueError("weight blocks: want %s of %s, got %s of %s" % (w_shape, w_dtype, tuple(wfrag.shape), wfrag.dtype))
        co = out_shape[1] * 8
        if bias.numel() != co or slope.numel() != co or bias.dtype != torch.float32 or slope.dtype != torch.float32:
            raise ValueError("bias / slope: want %d float32 values each" % co)
        slope01 = bool(tables[3]) if len(tables) > 3 else False
        out = torch.empty((n,) + out_shape, dtype=torch.float32, device=self.device)
        self._stream()
        check(fn(self.ctx, self._ptr(act), n, self._ptr(wfrag), self._ptr(bias), self._ptr(slope), 2 if slope01 else 0,
                 self._ptr(out)), self.ctx)
        return out

    def c3d2_conv41(self, act, tables):
        """svk_c3d2_conv41: chunked [n, 8 d, 8 chunks, 45, 8] (svk_c3d2_conv32t's output) -> conv4_1 + BN + PReLU
        -> chunked [n, 6 d, 16 chunks, 27 = 9 h x 3 w, 8]."""
        if tuple(act.shape[1:]) != (8, 8, 45, 8) or not act.is_contiguous():
            raise ValueError("c3d2_conv41 wants the chunked activation [n, 8, 8, 45, 8]")
        return self._c3d2_tail_conv(self.lib.svk_c3d2_conv41, act, tables, (6, 16, 27, 8), (8, 9, 2, 2, 64, 8), "float16")

    def c3d2_conv42(self, act, tables):
        """svk_c3d2_conv42: chunked [n, 6, 16, 27, 8] -> conv4_2 + BN + PReLU -> chunked [n, 4 d, 16 chunks, 9 = 3 h x 3 w, 8]."""
        if tuple(act.shape[1:]) != (6, 16, 27, 8) or not act.is_contiguous():
            raise ValueError("c3d2_conv42 wants the chunked activation [n, 6, 16, 27, 8]")
        return self._c3d2_tail_conv(self.lib.svk_c3d2_conv42, act, tables, (4, 16, 9, 8), (8, 16, 7, 4, 64, 2), "float32")

    def c3d2_fc5(self, act, tables):
        """svk_c3d2_fc5: chunked [n, 4, 16, 9, 8] (= [n, 4 608]) -> FC5 -> [n, 128] embeddings."""
        torch = _torch()
        n = act.shape[0]
        if act.numel() != n * 4608 or not act.is_contiguous():
            raise ValueError("c3d2_fc5 wants [n, 4608] (svk_c3d2_conv42's output)")
        wfrag, bias = tables
        work = torch.empty((int(self.lib.svk_c3d2_fc5_workspace_floats(n)),), dtype=torch.float32, device=self.device)
        out = torch.empty((n, 128), dtype=torch.float32, device=self.device)
        self._stream()
        check(self.lib.svk_c3d2_fc5(self.ctx, self._ptr(act), n, self._ptr(wfrag), self._ptr(bias), self._ptr(work),
                                    self._ptr(out)), self.ctx)
        return out

    def cosine_scores(self, test, enroll):
        torch = _torch()
        t = self.to_device(test, torch.float32)
        e = self.to_device(enroll, torch.float32)
        if t.dim() != 2 or e.dim() != 2 or t.shape[1] != e.shape[1]:
            raise ValueError("cosine_scores wants (Nt, D) and (Ns, D)")
        out = torch.empty((t.shape[0], e.shape[0]), dtype=torch.float32, device=self.device)
        self._stream()
        check(self.lib.svk_cosine_scores(self.ctx, self._ptr(t), self._ptr(e), t.shape[0], e.shape[0], t.shape[1],
                                         self._ptr(out)), self.ctx)
        return out

    def roc_eer(self, scores, labels):
        """(eer, auc) of flat scores / 0-1 labels, computed on the device (svk_roc_eer)."""
        torch = _torch()
        sc = self.to_device(scores, torch.float32).reshape(-1)
        lb = self.to_device(labels).reshape(-1)
        lb = (lb != 0).to(torch.uint8) if lb.dtype != torch.uint8 else lb
        n = sc.numel()
        if lb.numel() != n:
            raise ValueError("scores and labels differ in length")
        work = torch.empty((int(self.lib.svk_roc_workspace_bytes(n)),), dtype=torch.uint8, device=self.device)
        out = (C.c_double * 4)()
        self._stream()
        check(self.lib.svk_roc_eer(self.ctx, self._ptr(sc), self._ptr(lb), n, self._ptr(work), work.numel(), out),
              self.ctx)
        return float(out[0]), float(out[1])

    def l2_dist(self, a, b):
        torch = _torch()
        a = self.to_device(a, torch.float32)
        b = self.to_device(b, torch.float32)
        if a.shape != b.shape or a.dim() != 2:
            raise ValueError("l2_dist wants two (n, D) matrices")
        out = torch.empty((a.shape[0],), dtype=torch.float32, device=self.device)
        self._stream()
        check(self.lib.svk_l2_dist(self.ctx, self._ptr(a), self._ptr(b), a.shape[0], a.shape[1], self._ptr(out)),
              self.ctx)
        return out

    # ---- audio ingest ------------------------------------------------------------
    def resample(self, pcm, up, down, taps, lengths=None, out_dtype="f32"):
        """svk_ingest_resample: pcm int16 [n_utt, frames] or [n_utt, frames, channels] -> mono
        [n_utt, ceil(frames * up / down)] float32 in [-1, 1) or int16, plus the per-clip lengths."""
        torch = _torch()
        x = self.to_device(pcm)
        if x.dtype != torch.int16 or x.dim() not in (2, 3):
            raise ValueError("resample wants int16 PCM shaped (n_utt, frames) or (n_utt, frames, channels)")
        n_utt, n_in = int(x.shape[0]), int(x.shape[1])
        n_ch = int(x.shape[2]) if x.dim() == 3 else 1
        taps = np.asarray(taps, dtype=np.float32)
        key = (int(up), int(down), taps.size, float(taps[taps.size // 2]))
        cache = self.__dict__.setdefault("_resample_taps", {})
        if key not in cache:                         # the FIR is a per-ratio table: upload it once
            cache[key] = self.to_device(taps)
        h = cache[key]
        lens = self.to_device(np.asarray(lengths, dtype=np.int32)) if lengths is not None else None
        n_out = (n_in * up + down - 1) // down
        dt = {"f32": (torch.float32, _lib.PCM_F32), "i16": (torch.int16, _lib.PCM_I16)}[out_dtype]
        out = torch.empty((n_utt, max(n_out, 1)), dtype=dt[0], device=self.device)
        out_len = torch.empty((n_utt,), dtype=torch.int32, device=self.device)
        self._stream()
        check(self.lib.svk_ingest_resample(self.ctx, self._ptr(x), n_ch, n_in, self._ptr(lens), n_in, n_utt,
                                           self._ptr(h), int(h.numel()), int(up), int(down), self._ptr(out), dt[1],
                                           int(out.shape[1]), n_out, self._ptr(out_len)), self.ctx)
        return out[:, :n_out], out_len


_engines = {}


def get_engine(device=None):
    """Process-wide engine for `device` (default: torch's current CUDA device)."""
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("speaker_verification_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    index = torch.cuda.current_device() if device is None else int(device)
    eng = _engines.get(index)
    if eng is None:
        eng = _engines[index] = Engine(index)
    return eng
