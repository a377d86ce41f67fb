"""Audio ingest on the device: 16-bit WAV of any rate / channel count -> mono samples at
the model's 16 kHz, replacing the load step `librosa.load(path, sr=16000, mono=True)` of
`/root/reference/utils.py:170-173` (and feeding the int16 path of `vad.py:10-22`).

librosa is not installed and nothing in the reference pins its resampler, so the arithmetic is the
published polyphase scheme of `scipy.signal.resample_poly` (Kaiser-5 windowed sinc of half-width
10 * max(up, down), zero-padded edges) -- "parity unpinned" against librosa, pinned against SciPy in
`tests/`.  The taps are built here in float64 (host table, like the mel filterbank); the resampling
itself is `svk_ingest_resample` (csrc/ingest.hip).
"""
import contextlib
import math
import wave

import numpy as np

from . import constants as c


def rational_ratio(fs_in, fs_out):
    """(up, down) in lowest terms with fs_out / fs_in = up / down."""
    fs_in, fs_out = int(fs_in), int(fs_out)
    if fs_in <= 0 or fs_out <= 0:
        raise ValueError("sample rates must be positive")
    g = math.gcd(fs_in, fs_out)
    return fs_out // g, fs_in // g


def resample_taps(up, down, beta=5.0):
    """float64 FIR of scipy.signal.resample_poly's default design:
    `up * firwin(2 * half + 1, 1 / max(up, down), window=("kaiser", beta))`, half = 10 * max(up, down).
    firwin = windowed ideal low-pass, scaled to unit gain at DC."""
    max_rate = max(int(up), int(down))
    half = 10 * max_rate
    n = np.arange(-half, half + 1, dtype=np.float64)
    cutoff = 1.0 / max_rate                              # in units of the Nyquist frequency
    h = cutoff * np.sinc(cutoff * n) * np.kaiser(2 * half + 1, beta)
    h /= h.sum()
    return h * up


def read_wave_any(path):
    """(int16 frames [n_frames, n_channels], sample rate) of a 16-bit PCM WAV of any rate / channel
    count (the reference's `vad.read_wave` asserts mono and 8/16/32 kHz, vad.py:10-22)."""
    with contextlib.closing(wave.open(path, "rb")) as wf:
        if wf.getsampwidth() != 2:
            raise ValueError(f"{path}: need 16-bit PCM, got {8 * wf.getsampwidth()} bit")
        n_ch, rate = wf.getnchannels(), wf.getframerate()
        pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)
    return pcm.reshape(-1, n_ch), rate


def resample_batch(pcm, fs_in, fs_out=c.SAMPLE_RATE, lengths=None, out_dtype="f32", engine=None):
    """Batch of clips -> (mono samples at fs_out [n_utt, n_out] on the device, lengths [n_utt] int32).
    pcm: int16 [n_utt, frames] or [n_utt, frames, channels] (NumPy or device tensor)."""
    from .engine import get_engine
    eng = engine or get_engine()
    up, down = rational_ratio(fs_in, fs_out)
    return eng.resample(pcm, up, down, resample_taps(up, down), lengths=lengths, out_dtype=out_dtype)


def load_audio(path, sample_rate=c.SAMPLE_RATE):
    """float32 mono signal in [-1, 1) at `sample_rate`, like `librosa.load(path, sr=sample_rate,
    mono=True)[0]`: a mono file already at the rate is only rescaled (int16 / 32768, what librosa
    returns for it); anything else is down-mixed and resampled on the device."""
    frames, rate = read_wave_any(path)
    if rate == sample_rate and frames.shape[1] == 1:
        return frames[:, 0].astype(np.float32) / np.float32(32768.0)
    out, _ = resample_batch(frames[None], rate, sample_rate)
    return out[0].cpu().numpy()
