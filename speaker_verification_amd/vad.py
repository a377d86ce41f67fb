"""vad.py drop-in (`/root/reference/vad.py`): same `read_wave`, `write_wave`,
`Frame`, `frame_generator`, `vad_collector` call surface.

The reference asks `webrtcvad.Vad(3).is_speech(frame_bytes, sample_rate)` per
30 ms frame (vad.py:90,152) -- a third-party C extension that is neither
vendored nor installed.  The north-star replaces it with an energy rule owned
by this build: `EnergyVad(threshold)`.  When `vad_collector` is handed an
`EnergyVad`, framing + decision + hysteresis run in ONE gfx950 kernel
(`csrc/vad.hip`); any other object with an `is_speech` method is served by the
host-side protocol loop, because an opaque Python callback cannot run on a GPU.
`energy_vad_batch` is the batched entry point the pipeline uses.
"""
import collections
import contextlib
import sys
import wave

import numpy as np

from . import constants as c
from .engine import get_engine


def read_wave(path):
    """(PCM bytes, sample rate) of a mono 16-bit 8/16/32 kHz WAV (vad.py:10-22)."""
    with contextlib.closing(wave.open(path, 'rb')) as wf:
        assert wf.getnchannels() == 1
        assert wf.getsampwidth() == 2
        sample_rate = wf.getframerate()
        assert sample_rate in (8000, 16000, 32000)
        return wf.readframes(wf.getnframes()), sample_rate


def write_wave(path, audio, sample_rate):
    """Write PCM bytes as a mono 16-bit WAV (vad.py:25-33)."""
    with contextlib.closing(wave.open(path, 'wb')) as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(sample_rate)
        wf.writeframes(audio)


class Frame(object):
    """A frame of audio data (vad.py:36-41)."""

    def __init__(self, bytes, timestamp, duration):
        self.bytes = bytes
        self.timestamp = timestamp
        self.duration = duration


def frame_generator(frame_duration_ms, audio, sample_rate):
    """Successive `Frame`s of `frame_duration_ms` from PCM bytes; the loop
    condition is strict, so an exactly fitting last frame is dropped (Q12,
    vad.py:44-57)."""
    n = int(sample_rate * (frame_duration_ms / 1000.0) * 2)
    duration = (float(n) / sample_rate) / 2.0
    timestamp, offset = 0.0, 0
    while offset + n < len(audio):
        yield Frame(audio[offset:offset + n], timestamp, duration)
        timestamp += duration
        offset += n


class EnergyVad(object):
    """`is_speech(frame_bytes, sample_rate)`: sum(x^2) > threshold * n (int64)."""

    def __init__(self, threshold=c.VAD_ENERGY_THRESHOLD):
        self.threshold = int(threshold)

    def is_speech(self, frame_bytes, sample_rate):
        # single-frame form of the device rule (a ring of one frame: keep == speech flag);
        # one trailing pad sample because the framer needs offset + n < len (Q12)
        n = len(frame_bytes) // 2
        pcm = np.frombuffer(frame_bytes[:2 * n] + b"\0\0", dtype=np.int16)
        res = get_engine().vad_energy(pcm[None, :], self.threshold, fs=sample_rate, compact=False,
                                      frame_samples=n, ring_len=1)
        return bool(res["keep"][0, 0].item())

    def speech_flags(self, frames, sample_rate):
        """The decisions for ALL `frames` (equal-length `Frame`s, as `frame_generator` yields) with one
        launch: a caller that walks frames the reference's way (vad.py:90, one `is_speech` per 30 ms
        frame = one launch + one D2H each) should ask for them together."""
        frames = list(frames)
        if not frames:
            return np.zeros((0,), dtype=bool)
        n = len(frames[0].bytes) // 2
        pcm = np.frombuffer(b"".join(f.bytes[:2 * n] for f in frames) + b"\0\0", dtype=np.int16)
        # every frame is its own one-frame "clip" of n + 1 samples (the framer wants offset + n < len, Q12)
        # in the offsets / lengths form of the C-ABI; with a ring of one frame, keep == the speech flag
        count = len(frames)
        res = get_engine().vad_energy(pcm, self.threshold, fs=sample_rate, compact=False, frame_samples=n,
                                      ring_len=1, lengths=np.full((count,), n + 1, dtype=np.int32),
                                      offsets=np.arange(count, dtype=np.int64) * n)
        return res["keep"][:, 0].to("cpu").numpy().astype(bool)


def vad_collector(sample_rate, frame_duration_ms, padding_duration_ms, vad, frames):
    """Yield the voiced segments (bytes) of `frames` (vad.py:60-129): a ring
    buffer of padding/frame entries triggers when more than 90 % are voiced and
    releases when more than 90 % are unvoiced (Q13)."""
    frames = list(frames)
    if isinstance(vad, EnergyVad):
        if not frames:
            return
        n = len(frames[0].bytes)
        audio = b"".join(f.bytes for f in frames) + b"\0\0"      # the framer needs offset + n < len
        pcm = np.frombuffer(audio, dtype=np.int16)
        res = get_engine().vad_energy(pcm[None, :], vad.threshold, fs=sample_rate, frame_ms=frame_duration_ms,
                                      padding_ms=padding_duration_ms, compact=False, want_segments=True)
        seg = res["seg"][0].to("cpu").numpy()[:len(frames)]
        for k in range(int(seg.max()) + 1 if seg.size else 0):
            yield b"".join(frames[i].bytes for i in np.nonzero(seg == k)[0])
        return
    # opaque decision object: host-side protocol loop
    num_padding_frames = int(padding_duration_ms / frame_duration_ms)
    ring = collections.deque(maxlen=num_padding_frames)
    triggered, voiced = False, []
    # a decision object that can answer for all frames at once (one launch) is asked once
    flags = vad.speech_flags(frames, sample_rate) if hasattr(vad, "speech_flags") else None
    for k, frame in enumerate(frames):
        is_speech = bool(flags[k]) if flags is not None else vad.is_speech(frame.bytes, sample_rate)
        sys.stdout.write('1' if is_speech else '0')
        ring.append((frame, is_speech))
        if not triggered:
            if sum(1 for _, s in ring if s) > 0.9 * ring.maxlen:
                triggered = True
                voiced.extend(f for f, _ in ring)
                ring.clear()
        else:
            voiced.append(frame)
            if sum(1 for _, s in ring if not s) > 0.9 * ring.maxlen:
                triggered = False
                yield b''.join(f.bytes for f in voiced)
                ring.clear()
                voiced = []
    sys.stdout.write('\n')
    if voiced:
        yield b''.join(f.bytes for f in voiced)


def energy_vad_batch(pcm, threshold=c.VAD_ENERGY_THRESHOLD, sample_rate=c.SAMPLE_RATE,
                     frame_duration_ms=c.VAD_FRAME_MS, padding_duration_ms=c.VAD_PADDING_MS, lengths=None,
                     compact=True, want_segments=False):
    """[n_utt, L] int16 clips -> dict of device tensors: keep mask per 30 ms
    frame, frames per clip, voiced samples packed to the front of each row and
    their count (see Engine.vad_energy)."""
    return get_engine().vad_energy(pcm, threshold, fs=sample_rate, frame_ms=frame_duration_ms,
                                   padding_ms=padding_duration_ms, lengths=lengths, compact=compact,
                                   want_segments=want_segments)
