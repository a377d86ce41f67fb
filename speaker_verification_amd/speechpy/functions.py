"""speechpy.functions drop-in (`/root/reference/.../speechpy/functions.py`).

These four helpers are host-side table arithmetic (a few dozen scalars per
configuration), used to build the filterbank that is uploaded once per plan;
they stay in float64 NumPy on purpose: the filter edges hinge on float64 libm
rounding (Q2), so they are evaluated where the reference evaluates them.
"""
import numpy as np


def frequency_to_mel(f):
    """Hz -> mel (functions.py:26-32)."""
    return 1127 * np.log(1 + f / 700.)


def mel_to_frequency(mel):
    """mel -> Hz (functions.py:35-41)."""
    return 700 * (np.exp(mel / 1127.0) - 1)


def triangle(x, left, middle, right):
    """Unit-height triangle on [left, right] peaking at middle, sampled at x
    (functions.py:44-52)."""
    x = np.asarray(x, dtype=float)
    out = np.zeros(x.shape)
    up = (x > left) & (x <= middle)
    out[up] = (x[up] - left) / (middle - left)
    down = (x >= middle) & (x < right)          # evaluated second: x == middle takes this branch
    out[down] = (right - x[down]) / (right - middle)
    return out


def zero_handling(x):
    """Exact zeros -> float64 eps so that a log stays finite (functions.py:55-62)."""
    return np.where(x == 0, np.finfo(float).eps, x)
