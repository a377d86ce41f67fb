"""Drop-in for the reference's vendored SpeechPy 2.4
(`/root/reference/speech_feature_extraction/speechpy/`): same module names,
function names, positional order, keyword names and defaults.  The reference
ships the directory without an `__init__.py` (Q21); this one exports the three
sub-modules so `import speechpy; speechpy.feature.mfcc(...)` works.

Arithmetic runs on the GPU through libsvk.so; only shape bookkeeping and the
mel filterbank table (a host precomputation) are NumPy.
"""
from . import feature, functions, processing  # noqa: F401

__all__ = ["feature", "processing", "functions"]
