"""TEST INFRASTRUCTURE -- CPU restatement of the audio-ingest step, float64 NumPy.

The reference's call site is `librosa.load(filename, sr=16000, mono=True)` (/root/reference/utils.py:170-173).
librosa is absent from this image and unpinned in the reference (no requirements file), so its
resampler cannot be run or pinned: **parity unpinned** against librosa.  What is restated here is
the published algorithm the build uses instead, `scipy.signal.resample_poly(x, up, down)` with its
default design (SciPy 1.15.3: Kaiser beta 5 windowed sinc of half-width 10 * max(up, down), zero
padding at the edges), written out as the direct polyphase sum so the HIP kernel can be checked
line by line; `tests/test_oracle_golden.py` holds it to SciPy's own output.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np


def firwin_kaiser(up, down, beta=5.0):
    """up * firwin(2 * half + 1, 1 / max(up, down), window=('kaiser', beta)), half = 10 * max(up, down)."""
    max_rate = max(int(up), int(down))
    half = 10 * max_rate
    n = np.arange(-half, half + 1, dtype=np.float64)
    h = np.sinc(n / max_rate) / max_rate * np.kaiser(2 * half + 1, beta)
    return h / h.sum() * up


def to_mono(frames):
    """int16 [n_frames, n_channels] -> float64 mean over channels / 32768 (librosa: to_mono after the
    int16 -> float scaling of soundfile)."""
    frames = np.asarray(frames)
    if frames.ndim == 1:
        frames = frames[:, None]
    return frames.astype(np.float64).mean(axis=1) / 32768.0


def resample_poly(x, up, down, taps=None):
    """y[m] = sum_j x[j] * h[m * down + half - j * up], m < ceil(len(x) * up / down); zero outside x."""
    x = np.asarray(x, dtype=np.float64)
    h = firwin_kaiser(up, down) if taps is None else np.asarray(taps, dtype=np.float64)
    half = (len(h) - 1) // 2
    n_out = -(-len(x) * up // down)
    y = np.zeros(n_out, dtype=np.float64)
    m = np.arange(n_out, dtype=np.int64)
    t = m * down
    j_lo = -((half - t) // up)              # ceil((t - half) / up)
    j_hi = (t + half) // up
    for step in range(int((j_hi - j_lo).max()) + 1 if n_out else 0):
        j = j_lo + step
        ok = (j <= j_hi) & (j >= 0) & (j < len(x))
        k = t + half - j * up
        y[ok] += x[j[ok]] * h[k[ok]]
    return y


def to_int16(y):
    """back to the 16-bit grid: round half to even, saturating"""
    return np.clip(np.rint(np.asarray(y, dtype=np.float64) * 32768.0), -32768, 32767).astype(np.int16)
