"""speechpy.feature drop-in (`/root/reference/.../speechpy/feature.py`).

`mfcc`, `mfe`, `lmfe` keep the reference's signatures and float64 NumPy return
types and run the fused gfx950 front end (`csrc/frontend.hip`) on a batch of
one; `*_batch` variants (additive) take many clips and return device tensors.
`filterbanks` is the host-side table builder whose output is uploaded once per
configuration.
"""
import numpy as np

from . import functions, processing
from .. import _lib
from ..engine import get_engine, spec_from_seconds


def filterbanks(num_filter, coefficients, sampling_freq, low_freq=None, high_freq=None):
    """Mel filterbank, one triangular filter per row, columns = FFT bins
    (feature.py:33-99).  Keeps the reference's two quirks: a falsy `low_freq`
    (0 or None) becomes 300 Hz (Q1, feature.py:56) and band edges are mapped to
    bins with `(coefficients + 1) * hz / fs` (Q2, feature.py:77-82)."""
    high_freq = high_freq or sampling_freq / 2
    low_freq = low_freq or 300
    assert high_freq <= sampling_freq / 2, \
        "High frequency cannot be greater than half of the sampling frequency!"
    assert low_freq >= 0, "low frequency cannot be less than zero!"
    edges_mel = np.linspace(functions.frequency_to_mel(low_freq), functions.frequency_to_mel(high_freq),
                            num_filter + 2)
    edges_bin = np.floor((coefficients + 1) * functions.mel_to_frequency(edges_mel) / sampling_freq).astype(int)
    bank = np.zeros([num_filter, coefficients])
    for row, (lo, mid, hi) in enumerate(zip(edges_bin[:-2], edges_bin[1:-1], edges_bin[2:])):
        lo, mid, hi = int(lo), int(mid), int(hi)
        bank[row, lo:hi + 1] = functions.triangle(np.linspace(lo, hi, num=hi - lo + 1), left=lo, middle=mid,
                                                  right=hi)
    return bank


def _as_signal(signal):
    """1-D int16 / float32 view for the device; anything else is cast to float32
    (the reference casts to float64 at feature.py:182; the kernels compute in f32)."""
    signal = np.asarray(signal)
    if signal.ndim != 1:
        signal = signal.reshape(-1) if signal.ndim == 2 and 1 in signal.shape else signal
    assert signal.ndim == 1, \
        "Signal dimention should be of the format of (N,) but it is %s instead" % str(signal.shape)
    if signal.dtype == np.int16 or signal.dtype == np.float32:
        return np.ascontiguousarray(signal)
    return np.ascontiguousarray(signal, dtype=np.float32)


def _run_one(signal, sampling_frequency, frame_length, frame_stride, num_filters, fft_length, low_frequency,
             high_frequency, out_kind, num_cepstral=13, dc_elimination=True, want_energy=False):
    spec = spec_from_seconds(sampling_frequency, frame_length, frame_stride, fft_length, num_filters,
                             num_cepstral, out_kind, dc_elimination=dc_elimination, low_freq=low_frequency,
                             high_freq=high_frequency)
    sig = _as_signal(signal)
    n_frames = spec.num_frames(sig.shape[0])
    if n_frames <= 0:
        return np.empty((0, spec.num_cols)), np.empty((0,))
    # the fused kernel where it applies (fft_length 512 / 1024, <= 64 filters, bank within its bins);
    # Engine.features falls through to the staged kernels (framing -> spectrum -> mel / log / DCT)
    # for everything else, so every call the reference accepts computes (feature.py:77-99)
    feat, _, energy = get_engine().features(sig[None, :], spec, max_frames=n_frames, want_energy=want_energy)
    feat, energy = feat[0], (energy[0] if want_energy else None)
    out = feat.to("cpu").numpy().astype(np.float64)
    en = energy.to("cpu").numpy().astype(np.float64) if want_energy else None
    return out, en


def mfcc(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_cepstral=13, num_filters=40,
         fft_length=512, low_frequency=0, high_frequency=None, dc_elimination=True):
    """MFCC features, (num_frames, num_cepstral) float64 (feature.py:102-153)."""
    feat, _ = _run_one(signal, sampling_frequency, frame_length, frame_stride, num_filters, fft_length,
                       low_frequency, high_frequency, _lib.OUT_MFCC, num_cepstral, dc_elimination)
    if len(feat) == 0:
        return np.empty((0, num_cepstral))
    return feat


def mfe(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_filters=40, fft_length=512,
        low_frequency=0, high_frequency=None):
    """Mel filterbank energies and per-frame energies (feature.py:156-219)."""
    feat, energy = _run_one(signal, sampling_frequency, frame_length, frame_stride, num_filters, fft_length,
                            low_frequency, high_frequency, _lib.OUT_MFE, want_energy=True)
    return feat, energy


def lmfe(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_filters=40, fft_length=512,
         low_frequency=0, high_frequency=None):
    """Log mel filterbank energies (feature.py:222-258)."""
    feat, _ = _run_one(signal, sampling_frequency, frame_length, frame_stride, num_filters, fft_length,
                       low_frequency, high_frequency, _lib.OUT_LMFE)
    return feat


def extract_derivative_feature(feature):
    """(N, M) -> (N, M, 3): static, first and second derivative features
    (feature.py:261-282), both through `processing.derivative_extraction`."""
    first = processing.derivative_extraction(feature, DeltaWindows=2)
    second = processing.derivative_extraction(first, DeltaWindows=2)
    return np.concatenate((np.asarray(feature)[:, :, None], first[:, :, None], second[:, :, None]), axis=2)


# ---- additive, batched entry point ------------------------------------------------
def features_batch(pcm, sampling_frequency, kind="mfcc", frame_length=0.020, frame_stride=0.01, num_cepstral=13,
                   num_filters=40, fft_length=512, low_frequency=0, high_frequency=None, dc_elimination=True,
                   preemphasis_cof=None, lengths=None, want_energy=False):
    """Many clips in one launch.  pcm: [n_utt, L] int16/float32 (NumPy or CUDA
    tensor).  Returns device tensors (feat [n, T, C], n_frames [n], energy|None).
    `preemphasis_cof` fuses `processing.preemphasis(clip, cof=...)` in front."""
    out_kind = {"mfe": _lib.OUT_MFE, "lmfe": _lib.OUT_LMFE, "mfcc": _lib.OUT_MFCC}[kind]
    spec = spec_from_seconds(sampling_frequency, frame_length, frame_stride, fft_length, num_filters,
                             num_cepstral, out_kind, dc_elimination=dc_elimination, low_freq=low_frequency,
                             high_freq=high_frequency, preemph=preemphasis_cof is not None,
                             preemph_cof=preemphasis_cof if preemphasis_cof is not None else 0.0)
    return get_engine().features(pcm, spec, lengths=lengths, want_energy=want_energy)
