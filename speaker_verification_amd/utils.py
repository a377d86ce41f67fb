"""The transforms of `/root/reference/utils.py` that sit on the hot path, with the reference's
class names and sample-dict protocol (`{'feature': ..., 'label': ...}`), so a
`transforms.Compose([CMVN(), FeatureCube((80, 40, 20)), ToTensor()])` pipeline
(`utils.py:20-23`) keeps working: the array work runs in libsvk.so kernels.

Out of scope here: the dataset/file helpers, plotting, optimiser and
checkpoint plumbing of the reference's utils.py.
"""
import numpy as np

from . import constants as c
from .engine import get_engine
from .speechpy import feature as _feature, processing as _processing

np.random.seed(12345)          # the reference seeds the global NumPy RNG at import (utils.py:15, Q15)


class Compose(object):
    """`torchvision.transforms.Compose` (absent here): call the transforms in order."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample


def create_dataset(indexed_labels, origin_file_path):
    """`AudioDataset` over the WAVs under `constants.DATA_ORIGIN` listed in `origin_file_path`, with the
    transform chain CMVN -> FeatureCube((80, 40, 20)) -> ToTensor (utils.py:18-31)."""
    from .load_data import AudioDataset
    transform = Compose([CMVN(), FeatureCube((c.CUBE_FRAMES, c.NUM_COEF, c.CUBE_CROPS)), ToTensor()])
    return AudioDataset(origin_file_path, c.DATA_ORIGIN, indexed_labels=indexed_labels, transform=transform)


class ToTensor(object):
    """`(feature, label)` from a sample dict (utils.py:316-322)."""

    def __call__(self, sample):
        return sample['feature'], sample['label']


class FeatureCube(object):
    """`cube_shape = (num_frames, num_coefficient, num_utterances)`: `num_utterances` crops of
    `num_frames` consecutive feature rows at starts drawn from the GLOBAL NumPy RNG
    (`np.random.randint(T - num_frames, size=num_utterances)`, utils.py:372), stacked into a
    float32 `(1, num_utterances, num_frames, num_coefficient)` cube (utils.py:351-379)."""

    def __init__(self, cube_shape):
        assert isinstance(cube_shape, (tuple))
        self.cube_shape = cube_shape
        self.num_frames, self.num_coefficient, self.num_utterances = cube_shape[0], cube_shape[1], cube_shape[2]

    def __call__(self, sample):
        feature, label = np.asarray(sample['feature']), sample['label']
        idx = np.random.randint(feature.shape[0] - self.num_frames, size=self.num_utterances)
        cube = get_engine().cube_gather(feature[None].astype(np.float32), idx[None].astype(np.int32),
                                        self.num_frames)
        return {'feature': cube[0].to("cpu").numpy(), 'label': label}


class FeatureCube3C(object):
    """The 3-channel variant (static + two derivative channels, utils.py:325-348):
    feature (T, C, 3) -> cube (3, num_utterances, num_frames, C)."""

    def __init__(self, cube_shape):
        assert isinstance(cube_shape, (tuple))
        self.cube_shape = cube_shape
        self.num_frames, self.num_coefficient = cube_shape[0], cube_shape[1]
        self.num_utterances, self.num_channels = cube_shape[2], cube_shape[3]

    def __call__(self, sample):
        feature, label = np.asarray(sample['feature']).transpose(2, 0, 1), sample['label']   # (3, T, C)
        idx = np.random.randint(feature.shape[1] - self.num_frames, size=self.num_utterances)
        idx = np.tile(idx[None].astype(np.int32), (feature.shape[0], 1))                      # same crops per channel
        cube = get_engine().cube_gather(feature.astype(np.float32), idx, self.num_frames)    # (3, 1, U, F, C)
        return {'feature': cube[:, 0].to("cpu").numpy(), 'label': label}


class CMVN(object):
    """Optional derivative stacking and global CMVN, steered by `constants.DERIVATIVE` /
    `constants.NORMALIZE` exactly like utils.py:382-397 (identity with the shipped constants, Q16)."""

    def __call__(self, sample):
        feature, label = sample['feature'], sample['label']
        if c.DERIVATIVE:
            feature = _feature.extract_derivative_feature(feature)
            if c.NORMALIZE:
                for ch in range(3):
                    feature[:, :, ch] = _processing.cmvn(feature[:, :, ch], variance_normalization=True)
        elif c.NORMALIZE:
            feature = _processing.cmvn(feature, variance_normalization=True)
        return {'feature': feature, 'label': label}
