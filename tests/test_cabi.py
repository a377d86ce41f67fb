"""The C-ABI library builds, loads and exports exactly what include/svk.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    so = os.path.join(REPO, "speaker_verification_amd", "libsvk.so")
    if not os.path.exists(so):
        entry.build()
    from speaker_verification_amd import _lib
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "svk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svk_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    from speaker_verification_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 25
    assert sorted(_lib.SIGNATURES) == declared
    for name in declared:
        assert getattr(lib, name) is not None            # dlsym succeeds


def test_version_and_pure_host_entry_points(lib):
    from speaker_verification_amd._lib import FrontendCfg
    from speaker_verification_amd import _lib as binding
    header = open(os.path.join(REPO, "include", "svk.h")).read()
    assert lib.svk_version() == binding.VERSION == int(re.search(r"#define SVK_VERSION (\d+)", header).group(1))
    cfg = FrontendCfg(320, 160, 512, 40, 13, 2, 1, 0, 1, 0.98)
    assert lib.svk_frontend_num_frames(ctypes.byref(cfg), 48000) == 298       # Q3
    assert lib.svk_frontend_num_frames(ctypes.byref(cfg), 319) == 0
    assert lib.svk_frontend_num_cols(ctypes.byref(cfg)) == 13
    cfg.out_kind = 1
    assert lib.svk_frontend_num_cols(ctypes.byref(cfg)) == 40
    cfg_b = FrontendCfg(400, 160, 1024, 40, 40, 1, 1, 0, 1, 0.0)
    assert lib.svk_frontend_num_frames(ctypes.byref(cfg_b), 48000) == 297


def test_error_codes_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    handle = ctypes.c_void_p()
    assert lib.svk_create(0, ctypes.byref(handle)) == -4          # SVK_ERR_NO_DEVICE, no abort
    assert lib.svk_sync(None) == -1
    assert lib.svk_last_error(None) == b"null context"


def test_product_fails_loudly_without_gpu():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speaker_verification_amd.speechpy import feature
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        feature.mfcc(np.zeros(16000, dtype=np.int16), 16000)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(REPO, "speaker_verification_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
