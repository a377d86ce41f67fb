"""speaker_verification_amd -- MI355X-native feature -> embedding -> scoring path.

Drop-in call surface of MingmChen/Speaker_Verification for ONE hot path
(SURVEY.md section 8): `speechpy.{feature,processing,functions}`, `vad`,
`model.C3D2`, `siamese.Siamese`, `evaluation`.  Compute runs in hand-written
gfx950 HIP kernels behind the C-ABI of `include/svk.h` (`libsvk.so`, loaded by
`_lib.py` through ctypes), the whole C3D2 inference forward included (seven
MFMA kernels, reached through `model.C3D2.forward` itself in eval mode);
PyTorch-ROCm supplies device memory, streams, `torch.distributed` and the
training forward / backward.

Importing the package is cheap and needs no GPU; the first call into a device
op loads `libsvk.so` and raises if it is missing -- there is no CPU fallback.
"""
__all__ = ["speechpy", "vad", "model", "siamese", "evaluation", "pipeline", "distributed", "synth",
           "utils", "load_data", "ingest", "engine", "constants", "train_siamese"]
