"""CPU oracle for the VAD chunker -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates `vad.py` of the reference (paths relative to `/root/reference/`):
the 30 ms framer (`vad.py:44-57`, Q12) and the ring-buffer hysteresis
(`vad.py:60-129`, Q13).

The per-frame speech decision in the reference is `webrtcvad.Vad(3).is_speech`
(`vad.py:90,152`), a third-party C extension (py-webrtcvad, version unpinned --
the reference has no requirements file) that is absent from the reference tree
and from this image.  The north-star replaces it with an ENERGY rule owned by
this build:

    is_speech(frame) := sum(x[i]^2) > threshold * n_samples      (all int64)

Parity status: the framer and the hysteresis are PINNED by golden masks that
`tools/make_golden.py` produced by running the reference's own
`frame_generator` / `vad_collector` with this energy rule plugged in as the
`vad` object.  The decision rule itself is "parity unpinned" against webrtcvad
(nothing in the reference pins it); "bit-exact" is between this oracle and the
HIP kernel.
"""
import collections

import numpy as np


def frame_bytes(frame_duration_ms, sample_rate):
    """Bytes per frame, `int(sr * (ms / 1000.0) * 2)`.  vad.py:50."""
    return int(sample_rate * (frame_duration_ms / 1000.0) * 2)


def num_frames(n_audio_bytes, frame_duration_ms, sample_rate):
    """How many frames `frame_generator` yields: it loops while
    `offset + n < len(audio)` (strict, Q12), so an exactly fitting last frame is
    dropped.  vad.py:50-57."""
    n = frame_bytes(frame_duration_ms, sample_rate)
    count, offset = 0, 0
    while offset + n < n_audio_bytes:
        count += 1
        offset += n
    return count


def energy_is_speech(pcm_frame, threshold):
    """The build's integer decision rule on one frame of int16 samples."""
    x = np.asarray(pcm_frame, dtype=np.int64)
    return bool(int(np.sum(x * x)) > int(threshold) * x.shape[0])


def frame_flags(pcm, frame_duration_ms, sample_rate, threshold):
    """Per-frame speech booleans for an int16 clip."""
    pcm = np.asarray(pcm, dtype=np.int16)
    n = frame_bytes(frame_duration_ms, sample_rate) // 2
    nf = num_frames(pcm.shape[0] * 2, frame_duration_ms, sample_rate)
    return np.array([energy_is_speech(pcm[f * n:(f + 1) * n], threshold)
                     for f in range(nf)], dtype=bool)


def collect(flags, frame_duration_ms=30, padding_duration_ms=300):
    """Hysteresis of vad.py:81-129 on a sequence of per-frame booleans.

    Returns (keep, seg): keep[f] is True when frame f is part of some yielded
    segment; seg[f] is that segment's ordinal (or -1).  A segment is what one
    `yield` of the reference's generator concatenates.
    """
    maxlen = int(padding_duration_ms / frame_duration_ms)          # vad.py:81
    ring = collections.deque(maxlen=maxlen)
    keep = np.zeros(len(flags), dtype=bool)
    seg = np.full(len(flags), -1, dtype=np.int32)
    triggered = False
    cur = 0
    for f, speech in enumerate(flags):
        if not triggered:
            ring.append((f, bool(speech)))
            voiced = sum(1 for _, s in ring if s)
            if voiced > 0.9 * ring.maxlen:                           # vad.py:99
                triggered = True
                for g, _ in ring:                                    # vad.py:105-106
                    keep[g] = True
                    seg[g] = cur
                ring.clear()
        else:
            keep[f] = True                                           # vad.py:111
            seg[f] = cur
            ring.append((f, bool(speech)))
            unvoiced = sum(1 for _, s in ring if not s)
            if unvoiced > 0.9 * ring.maxlen:                         # vad.py:117
                triggered = False
                cur += 1
                ring.clear()
    return keep, seg


def vad_energy(pcm, sample_rate=16000, frame_duration_ms=30,
               padding_duration_ms=300, threshold=250000):
    """Full chain on one int16 clip: flags -> hysteresis.  Returns
    (keep mask, segment ids, voiced samples concatenated in order)."""
    pcm = np.asarray(pcm, dtype=np.int16)
    flags = frame_flags(pcm, frame_duration_ms, sample_rate, threshold)
    keep, seg = collect(flags, frame_duration_ms, padding_duration_ms)
    n = frame_bytes(frame_duration_ms, sample_rate) // 2
    parts = [pcm[f * n:(f + 1) * n] for f in range(len(keep)) if keep[f]]
    voiced = np.concatenate(parts) if parts else np.zeros((0,), dtype=np.int16)
    return keep, seg, voiced
