"""`AudioDataset` with the constructor and item protocol of `/root/reference/load_data.py:9-87`
(thin host plumbing on the path: file list -> WAV -> log-mel features -> transforms).

The reference reads audio with `librosa.load(path, sr=16000, mono=True)` (`utils.py:170-173`), which is
not installed here.  `load_wav` reads 16-bit PCM WAV files with the standard library and returns what
librosa does: float32 mono in [-1, 1) at the target rate.  A mono file already at the rate is only
rescaled (`int16 / 32768`); other rates / channel counts are down-mixed and resampled on the device
(`ingest.py`, `svk_ingest_resample`) with SciPy's published polyphase design -- librosa's own resampler
is third-party arithmetic that nothing in the reference pins ("parity unpinned" for that case).
"""
import os

import numpy as np

from . import constants as c
from .ingest import load_audio
from .speechpy import feature as speech
from .utils import ToTensor


def load_wav(filename, sample_rate=c.SAMPLE_RATE):
    """float32 mono signal in [-1, 1) at `sample_rate` of a 16-bit PCM WAV (utils.py:170-173)."""
    return load_audio(filename, sample_rate)


class AudioDataset(object):
    """files_path: text file with one relative WAV path per line; audio_dir: their root;
    indexed_labels: {first 7 characters of the path (the speaker id): class index}; transform: callable
    on `{'feature', 'label'}` (load_data.py:10-45).  Files that are missing or smaller than 1 000 bytes
    are skipped like the reference does (load_data.py:28-39)."""

    def __init__(self, files_path, audio_dir, indexed_labels, transform=None, derivative=c.DERIVATIVE):
        self.audio_dir = audio_dir
        self.transform = transform
        self.indexed = indexed_labels
        self.derivative = derivative
        content = np.atleast_1d(np.genfromtxt(files_path, dtype='str'))
        kept = []
        for rel in content:
            full = os.path.join(self.audio_dir, rel)
            try:
                assert os.path.getsize(full) > 1000, "Bad file!"
                kept.append(str(rel))
            except OSError as err:
                print("OS error: {0}".format(err))
        self.sound_files = kept if c.NUM_FILES == 0 else kept[:c.NUM_FILES]      # load_data.py:42-45

    def __len__(self):
        return len(self.sound_files)

    def load_signal(self, idx):
        return load_wav(os.path.join(self.audio_dir, self.sound_files[idx]))

    def __getitem__(self, idx):
        signal = self.load_signal(idx)
        logenergy = speech.lmfe(signal, sampling_frequency=c.SAMPLE_RATE, frame_length=c.FRAME_LEN,
                                frame_stride=c.FRAME_STEP, num_filters=c.NUM_COEF, fft_length=c.NUM_FFT)
        sample = {'feature': logenergy, 'label': self.indexed[self.sound_files[idx][0:7]]}   # load_data.py:73
        if self.transform:
            return self.transform(sample)
        return ToTensor()(sample)
