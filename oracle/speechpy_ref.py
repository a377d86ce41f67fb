"""CPU oracle for the SpeechPy front end -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A float64 NumPy restatement of the algorithms in the reference's vendored
SpeechPy 2.4 (`speech_feature_extraction/speechpy/{processing,feature,functions}.py`).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product package never does.

Parity status: PINNED.  `tools/make_golden.py` imported the reference in the
build container (numpy 2.2.6 / scipy 1.15.3) and dumped `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against those vectors
to <= 1e-12.

Every function cites the reference lines it follows (paths relative to
`/root/reference/speech_feature_extraction/speechpy/`).  The reference quirks
that matter for parity are listed in SURVEY.md section 0.1 (Q1..Q11) and are
reproduced on purpose.
"""
import math

import numpy as np

_EPS64 = float(np.finfo(np.float64).eps)


# --------------------------------------------------------------------------
# functions.py
# --------------------------------------------------------------------------
def frequency_to_mel(f):
    """Hz -> mel, 1127 ln(1 + f/700).  functions.py:26-32."""
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_frequency(mel):
    """mel -> Hz, 700 (exp(mel/1127) - 1).  functions.py:35-41."""
    return 700.0 * (np.exp(np.asarray(mel, dtype=np.float64) / 1127.0) - 1.0)


def triangle(x, left, middle, right):
    """Triangular window sampled at `x`: rises on (left, middle], falls on
    [middle, right), zero outside.  functions.py:44-52 (the falling-edge
    assignment runs second, so x == middle takes (right-x)/(right-middle))."""
    x = np.asarray(x, dtype=np.float64)
    y = np.zeros(x.shape)
    up = (x > left) & (x <= middle)
    y[up] = (x[up] - left) / (middle - left)
    down = (x >= middle) & (x < right)
    y[down] = (right - x[down]) / (right - middle)
    return y


def zero_handling(x):
    """Exact zeros become float64 eps (so a later log is finite).
    functions.py:55-62."""
    x = np.asarray(x)
    return np.where(x == 0, _EPS64, x)


# --------------------------------------------------------------------------
# processing.py
# --------------------------------------------------------------------------
def preemphasis(signal, shift=1, cof=0.98):
    """y[n] = x[n] - cof * x[(n - shift) mod N]  (circular, Q5).
    processing.py:45-58.  Dtype follows NumPy promotion: int16 -> float64,
    float32 -> float32."""
    signal = np.asarray(signal)
    return signal - cof * np.roll(signal, shift)


def frame_geometry(length_signal, sampling_frequency, frame_length, frame_stride,
                   zero_padding):
    """(samples per frame, stride in samples, number of frames) exactly as
    processing.py:93-117 derives them (Q3: the no-padding branch has no '+1')."""
    flen = int(np.round(sampling_frequency * frame_length))
    stride = float(np.round(sampling_frequency * frame_stride))
    span = (length_signal - flen) / stride
    nframes = int(math.ceil(span)) if zero_padding else int(math.floor(span))
    return flen, stride, nframes


def stack_frames(sig, sampling_frequency, frame_length=0.020, frame_stride=0.020,
                 filter=lambda x: np.ones((x,)), zero_padding=True):
    """Overlapping frames of `sig`, each multiplied by the window `filter(flen)`.
    processing.py:61-139."""
    sig = np.asarray(sig)
    assert sig.ndim == 1, \
        "Signal dimention should be of the format of (N,) but it is %s instead" % str(sig.shape)
    flen, stride, nframes = frame_geometry(sig.shape[0], sampling_frequency,
                                           frame_length, frame_stride, zero_padding)
    if zero_padding:
        total = int(nframes * stride + flen)                       # :107-109
        padded = np.concatenate((sig, np.zeros((total - sig.shape[0],))))
    else:
        total = int((nframes - 1) * stride + flen)                 # :119-120
        padded = sig[0:total]
    # start of frame t is t*stride (float), truncated to int32 with the offset
    # already added -- same order of operations as processing.py:123-131.
    starts = np.arange(0, nframes * stride, stride)
    idx = (starts[:, None] + np.arange(0, flen)[None, :]).astype(np.int32)
    frames = padded[idx]
    return frames * np.asarray(filter(flen))[None, :]              # :137-138


def fft_spectrum(frames, fft_points=512):
    """|rfft(frame, n=fft_points)| per row.  processing.py:142-159."""
    return np.absolute(np.fft.rfft(frames, n=fft_points, axis=-1, norm=None))


def power_spectrum(frames, fft_points=512):
    """(1/N) |rfft|^2 (Q6).  processing.py:162-174."""
    return 1.0 / fft_points * np.square(fft_spectrum(frames, fft_points))


def log_power_spectrum(frames, fft_points=512, normalize=True):
    """10 log10 of the power spectrum floored at 1e-20, optionally shifted so
    its global maximum is 0 dB.  processing.py:177-198."""
    p = power_spectrum(frames, fft_points)
    p = np.where(p <= 1e-20, 1e-20, p)
    lp = 10.0 * np.log10(p)
    return lp - np.max(lp) if normalize else lp


def derivative_extraction(feat, DeltaWindows):
    """'Delta' features as the reference actually computes them (Q11): the
    subtraction on processing.py:232 is a detached expression statement, so
    the result is sum_r r * FEAT[:, D+r : D+r+cols] / sum_r 2 r^2, with edge
    padding along the FEATURE axis (processing.py:223)."""
    feat = np.asarray(feat)
    rows, cols = feat.shape
    padded = np.pad(feat, ((0, 0), (DeltaWindows, DeltaWindows)), 'edge')
    acc = np.zeros(feat.shape, dtype=feat.dtype)
    scale = 0
    for r in range(1, DeltaWindows + 1):
        acc += r * padded[:, DeltaWindows + r:DeltaWindows + r + cols]
        scale += 2 * r * r
    return acc / scale


def cmvn(vec, variance_normalization=False):
    """Per-column mean removal over rows; optionally divide by (population std
    + 2^-30) (Q9).  processing.py:239-271."""
    vec = np.asarray(vec)
    centred = vec - np.mean(vec, axis=0)[None, :]
    if not variance_normalization:
        return centred
    return centred / (np.std(centred, axis=0)[None, :] + 2.0 ** -30)


def cmvnw(vec, win_size=301, variance_normalization=False):
    """Sliding-window CMVN (Q10): 'symmetric' padding of (win-1)/2 rows, float32
    output; the variance pass windows over the mean-subtracted float32 array.
    processing.py:274-327."""
    vec = np.asarray(vec)
    rows, cols = vec.shape
    assert isinstance(win_size, int), "Size must be of type 'int'!"
    assert win_size % 2 == 1, "Windows size must be odd!"
    half = int((win_size - 1) / 2)
    padded = np.pad(vec, ((half, half), (0, 0)), 'symmetric')
    centred = np.zeros(vec.shape, dtype=np.float32)
    for i in range(rows):
        centred[i] = vec[i] - np.mean(padded[i:i + win_size], axis=0)
    if not variance_normalization:
        return centred
    padded2 = np.pad(centred, ((half, half), (0, 0)), 'symmetric')
    out = np.zeros(vec.shape, dtype=np.float32)
    for i in range(rows):
        out[i] = centred[i] / (np.std(padded2[i:i + win_size], axis=0) + 2.0 ** -30)
    return out


# --------------------------------------------------------------------------
# feature.py
# --------------------------------------------------------------------------
def mel_edges(num_filter, coefficients, sampling_freq, low_freq=None, high_freq=None):
    """Integer FFT-bin index of the num_filter+2 mel-spaced band edges.
    feature.py:55-82, including Q1 (`low_freq or 300`) and Q2
    (`(coefficients + 1) * hz / fs` with coefficients = nfft//2 + 1)."""
    high_freq = high_freq or sampling_freq / 2
    low_freq = low_freq or 300
    assert high_freq <= sampling_freq / 2, \
        "High frequency cannot be greater than half of the sampling frequency!"
    assert low_freq >= 0, "low frequency cannot be less than zero!"
    mels = np.linspace(frequency_to_mel(low_freq), frequency_to_mel(high_freq),
                       num_filter + 2)
    hertz = mel_to_frequency(mels)
    return np.floor((coefficients + 1) * hertz / sampling_freq).astype(int)


def filterbanks(num_filter, coefficients, sampling_freq, low_freq=None, high_freq=None):
    """(num_filter, coefficients) triangular mel filterbank.  feature.py:33-99."""
    edges = mel_edges(num_filter, coefficients, sampling_freq, low_freq, high_freq)
    bank = np.zeros([num_filter, coefficients])
    for i in range(num_filter):
        left, middle, right = int(edges[i]), int(edges[i + 1]), int(edges[i + 2])
        bins = np.linspace(left, right, num=right - left + 1)
        bank[i, left:right + 1] = triangle(bins, left=left, middle=middle, right=right)
    return bank


def mfe(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01,
        num_filters=40, fft_length=512, low_frequency=0, high_frequency=None):
    """Mel filterbank energies and per-frame total energy.  feature.py:156-219:
    float64 cast, rectangular framing without padding (Q4), power spectrum,
    energy = sum over ALL bins, zero -> eps on both outputs (Q7)."""
    signal = np.asarray(signal).astype(float)
    frames = stack_frames(signal, sampling_frequency=sampling_frequency,
                          frame_length=frame_length, frame_stride=frame_stride,
                          filter=lambda x: np.ones((x,)), zero_padding=False)
    high_frequency = high_frequency or sampling_frequency / 2
    power = power_spectrum(frames, fft_length)
    energies = zero_handling(np.sum(power, 1))
    bank = filterbanks(num_filters, power.shape[1], sampling_frequency,
                       low_frequency, high_frequency)
    feats = zero_handling(np.dot(power, bank.T))
    return feats, energies


def lmfe(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01,
         num_filters=40, fft_length=512, low_frequency=0, high_frequency=None):
    """log of mfe features.  feature.py:222-258."""
    feats, _ = mfe(signal, sampling_frequency, frame_length, frame_stride,
                   num_filters, fft_length, low_frequency, high_frequency)
    return np.log(feats)


def dct2_ortho_matrix(n_out, n_in):
    """First n_out rows of the orthonormal DCT-II matrix of size n_in: what
    scipy.fftpack.dct(type=2, norm='ortho') multiplies by (feature.py:147)."""
    k = np.arange(n_out)[:, None]
    n = np.arange(n_in)[None, :]
    mat = np.sqrt(2.0 / n_in) * np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in))
    mat[0, :] = np.sqrt(1.0 / n_in)
    return mat


def mfcc(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01,
         num_cepstral=13, num_filters=40, fft_length=512, low_frequency=0,
         high_frequency=None, dc_elimination=True):
    """MFCC.  feature.py:102-153: log mel energies -> DCT-II (ortho) along the
    filter axis -> first num_cepstral -> column 0 := log(frame energy) (Q8)."""
    feats, energies = mfe(signal, sampling_frequency, frame_length, frame_stride,
                          num_filters, fft_length, low_frequency, high_frequency)
    if len(feats) == 0:
        return np.empty((0, num_cepstral))
    ceps = np.log(feats) @ dct2_ortho_matrix(num_cepstral, feats.shape[1]).T
    if dc_elimination:
        ceps[:, 0] = np.log(energies)
    return ceps


def extract_derivative_feature(feature):
    """(N, M) -> (N, M, 3): static, first and second 'derivative' (window 2).
    feature.py:261-282."""
    d1 = derivative_extraction(feature, DeltaWindows=2)
    d2 = derivative_extraction(d1, DeltaWindows=2)
    return np.stack((feature, d1, d2), axis=2)
