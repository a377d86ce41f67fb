"""speechpy.processing drop-in (`/root/reference/.../speechpy/processing.py`).

Same names, argument order, keyword names, defaults and assertions.  Each
function ships its array to the GPU, runs the matching libsvk.so kernel
(`csrc/stages.hip`) and returns NumPy in the reference's dtype.  The kernels
compute in float32; the reference computes in float64 -- results agree to the
tolerances stated in tests/test_gpu_parity.py.
"""
import decimal
import math

import numpy as np

from ..engine import get_engine


def round_half_up(number):
    """1.4 -> 1, 1.5 -> 2, 1.6 -> 2 (processing.py:38-42; unused by the path)."""
    return int(decimal.Decimal(number).quantize(decimal.Decimal('1'), rounding=decimal.ROUND_HALF_UP))


def _host(t, dtype=np.float64):
    return t.to("cpu").numpy().astype(dtype)


def preemphasis(signal, shift=1, cof=0.98):
    """y[n] = x[n] - cof * x[(n - shift) mod N], circular like np.roll
    (processing.py:45-58).  float32 in -> float32 out, otherwise float64 out."""
    signal = np.asarray(signal)
    if signal.size == 0:
        return signal - cof * signal
    flat = signal.reshape(-1)                   # np.roll without axis rolls the flattened array
    if flat.dtype != np.int16 and flat.dtype != np.float32:
        flat = flat.astype(np.float32)
    out = get_engine().preemphasis(flat, shift=shift, cof=cof)
    out_dtype = np.float32 if signal.dtype == np.float32 else np.float64
    return _host(out, out_dtype).reshape(signal.shape)


def stack_frames(sig, sampling_frequency, frame_length=0.020, frame_stride=0.020,
                 filter=lambda x: np.ones((x,)), zero_padding=True):
    """Frame a signal into overlapping frames (processing.py:61-139).  The
    window `filter(frame_len)` is evaluated on the host and applied on device."""
    sig = np.asarray(sig)
    s = "Signal dimention should be of the format of (N,) but it is %s instead"
    assert sig.ndim == 1, s % str(sig.shape)
    length_signal = sig.shape[0]
    frame_sample_length = int(np.round(sampling_frequency * frame_length))
    frame_stride = float(np.round(sampling_frequency * frame_stride))
    span = (length_signal - frame_sample_length) / frame_stride
    numframes = int(math.ceil(span)) if zero_padding else int(math.floor(span))
    if numframes <= 0:
        return np.zeros((0, frame_sample_length))
    window = np.asarray(filter(frame_sample_length), dtype=np.float64)
    if zero_padding:
        usable = length_signal                       # zeros are supplied past the end (processing.py:107-109)
    else:
        usable = int((numframes - 1) * frame_stride + frame_sample_length)   # processing.py:119-120
    frames = get_engine().stack_frames(sig[:usable], frame_sample_length, int(frame_stride), numframes,
                                       None if np.all(window == 1.0) else window)
    return _host(frames)


def fft_spectrum(frames, fft_points=512):
    """|rfft(frame, n=fft_points)| per row (processing.py:142-159)."""
    return _host(get_engine().spectrum(np.asarray(frames), fft_points, power=False))


def power_spectrum(frames, fft_points=512):
    """(1 / fft_points) |rfft|^2 per row (processing.py:162-174)."""
    return _host(get_engine().spectrum(np.asarray(frames), fft_points, power=True))


def log_power_spectrum(frames, fft_points=512, normalize=True):
    """10 log10 of the power spectrum floored at 1e-20; with `normalize` the
    global maximum is shifted to 0 dB (processing.py:177-198)."""
    eng = get_engine()
    p = eng.spectrum(np.asarray(frames), fft_points, power=True)
    return _host(eng.log_power_(p, normalize=normalize))


def derivative_extraction(feat, DeltaWindows):
    """Derivative features exactly as the reference computes them, bug included
    (Q11, processing.py:201-236): edge-pad DeltaWindows columns on both sides of
    the FEATURE axis; DIF = sum_r r * FEAT[:, D + r : D + r + cols]; / sum_r 2 r^2."""
    feat = np.asarray(feat)
    rows, cols = feat.shape
    return _host(get_engine().derivative(feat, DeltaWindows))


def cmvn(vec, variance_normalization=False):
    """Global cepstral mean (and variance) normalisation, one observation per
    row (processing.py:239-271)."""
    import torch
    eng = get_engine()
    vec = np.asarray(vec)
    rows, cols = vec.shape
    x = eng.to_device(vec, torch.float32).clone()
    eng.cmvn_(x, variance=variance_normalization)
    return _host(x)


def cmvnw(vec, win_size=301, variance_normalization=False):
    """Sliding-window CMVN, float32 output (Q10, processing.py:274-327):
    'symmetric' padding of (win_size - 1) / 2 rows, window mean removed; the
    variance pass windows over the mean-subtracted array padded the same way."""
    vec = np.asarray(vec)
    rows, cols = vec.shape
    assert isinstance(win_size, int), "Size must be of type 'int'!"
    assert win_size % 2 == 1, "Windows size must be odd!"
    out = get_engine().cmvnw(vec, win_size=win_size, variance=variance_normalization)
    return _host(out, np.float32)
