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


def main(id_list=None, chunked_list='100_speakers_100_samples_chunked_ids.txt', threshold=c.VAD_ENERGY_THRESHOLD, batch=512):
    """vad.py:135-168, the file-driven chunker: every WAV named in the id list (`c.ROOT/100_first_ids_100_samples.txt`,
    the list the reference's second assignment leaves in effect) is read from `c.DATA_ORIGIN/wav/`, cut into its voiced
    segments (30 ms frames, 300 ms of padding) and written as `c.DATA_ORIGIN/wav_chunked/<name>_<i>.wav`; the new
    relative names go to `chunked_list` in the working directory.  The reference walks file by file and frame by frame
    through webrtcvad; here `batch` files share ONE ragged launch of the energy VAD (offsets / lengths into one buffer)
    and the host only slices the segments out.  Returns the list of written names."""
    import os
    id_list = id_list or os.path.join(c.ROOT, '100_first_ids_100_samples.txt')
    train_files = [str(f) for f in np.atleast_1d(np.genfromtxt(id_list, dtype='str'))]
    to_path = os.path.join(c.DATA_ORIGIN, 'wav_chunked')
    from_path = os.path.join(c.DATA_ORIGIN, 'wav')
    os.makedirs(to_path, exist_ok=True)
    train_list = []
    eng = get_engine()
    for lo in range(0, len(train_files), batch):
        names = train_files[lo:lo + batch]
        audio = [read_wave(os.path.join(from_path, f)) for f in names]
        for rate in sorted({r for _, r in audio}):                    # one launch per sample rate present
            idx = [k for k, (_, r) in enumerate(audio) if r == rate]
            pcm = [np.frombuffer(audio[k][0], dtype=np.int16) for k in idx]
            lens = np.array([x.size for x in pcm], dtype=np.int32)
            slots = (lens.astype(np.int64) + 7) // 8 * 8
            offs = np.concatenate([[0], np.cumsum(slots)[:-1]]).astype(np.int64)
            buf = np.zeros((int(slots.sum()) if len(idx) else 0,), dtype=np.int16)
            for x, o in zip(pcm, offs):
                buf[o:o + x.size] = x
            res = eng.vad_energy(buf, threshold, fs=rate, frame_ms=c.VAD_FRAME_MS, padding_ms=c.VAD_PADDING_MS,
                                 lengths=lens, offsets=offs, compact=False, want_segments=True)
            seg = res["seg"].to("cpu").numpy()
            n = res["frame_samples"]
            for row, k in enumerate(idx):
                nseg = int(seg[row].max()) + 1
                for i in range(nseg):
                    frames = np.nonzero(seg[row] == i)[0]
                    segment = b"".join(pcm[row][f * n:(f + 1) * n].tobytes() for f in frames)
                    dest_path = os.path.join(to_path, names[k]).replace('.wav', '_{}.wav'.format(i))
                    train_list.append(dest_path.replace(to_path + '/', ''))
                    os.makedirs(os.path.dirname(dest_path), exist_ok=True)
                    write_wave(dest_path, segment, rate)
    # (file order inside a batch follows the sample rates present; the reference's list is in file order)
    order = {f: k for k, f in enumerate(train_files)}
    train_list.sort(key=lambda name: (order[name.rsplit('_', 1)[0] + '.wav'], int(name.rsplit('_', 1)[1][:-4])))
    np.savetxt(chunked_list, train_list, fmt='%s')
    return train_list


if __name__ == '__main__':
    main()
