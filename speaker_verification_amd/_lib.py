"""ctypes binding of libsvk.so (C-ABI: include/svk.h).

This is the only place that touches the shared library.  There is NO fallback:
if `libsvk.so` is missing or a call fails, an exception is raised.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C speaker_verification_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvk.so")

SVK_OK = 0
VERSION = 111                      # include/svk.h SVK_VERSION
SVK_ERR_BAD_ARG, SVK_ERR_UNSUPPORTED, SVK_ERR_HIP, SVK_ERR_NO_DEVICE, SVK_ERR_OOM, SVK_ERR_RCCL = -1, -2, -3, -4, -5, -6
OUT_MFE, OUT_LMFE, OUT_MFCC = 0, 1, 2
PCM_I16, PCM_F32 = 0, 1

_STATUS_NAMES = {-1: "SVK_ERR_BAD_ARG", -2: "SVK_ERR_UNSUPPORTED", -3: "SVK_ERR_HIP",
                 -4: "SVK_ERR_NO_DEVICE", -5: "SVK_ERR_OOM", -6: "SVK_ERR_RCCL"}


class SvkError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{_STATUS_NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


class FrontendCfg(C.Structure):
    """mirror of struct svk_frontend_cfg"""
    _fields_ = [("frame_len", C.c_int32), ("frame_stride", C.c_int32), ("nfft", C.c_int32),
                ("num_filters", C.c_int32), ("num_ceps", C.c_int32), ("out_kind", C.c_int32),
                ("dc_elimination", C.c_int32), ("preemph", C.c_int32), ("preemph_shift", C.c_int32),
                ("preemph_cof", C.c_float), ("input_scale", C.c_float)]


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes).  Must list every symbol include/svk.h declares
# (tests/test_cabi.py checks the two against each other).
SIGNATURES = {
    "svk_version": (C.c_int, []),
    "svk_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "svk_destroy": (None, [_vp]),
    "svk_last_error": (C.c_char_p, [_vp]),
    "svk_set_stream": (C.c_int, [_vp, _vp]),
    "svk_sync": (C.c_int, [_vp]),
    "svk_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "svk_free": (C.c_int, [_vp, _vp]),
    "svk_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "svk_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "svk_memset": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t]),
    "svk_device_info": (C.c_int, [_vp, C.POINTER(_i64)]),
    "svk_frontend_plan_create": (C.c_int, [_vp, C.POINTER(FrontendCfg), C.POINTER(C.c_double), C.POINTER(_vp)]),
    "svk_frontend_plan_destroy": (None, [_vp]),
    "svk_frontend_num_frames": (_i64, [C.POINTER(FrontendCfg), _i64]),
    "svk_frontend_num_cols": (C.c_int, [C.POINTER(FrontendCfg)]),
    "svk_frontend_run": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32]),
    "svk_preemphasis": (C.c_int, [_vp, _vp, C.c_int, _i64, _i32, _f32, _vp]),
    "svk_stack_frames": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "svk_spectrum": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "svk_cmvn": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _i32]),
    "svk_cmvn_stats": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "svk_mel_features": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "svk_cmvnw": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "svk_derivative": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "svk_log_power": (C.c_int, [_vp, _vp, _i64, _i32]),
    "svk_vad_energy": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i64, _i32,
                                  _vp, _vp, _vp, _vp, _vp, _vp]),
    "svk_cube_draw_crops": (C.c_int, [_vp, _vp, _i32, _i64, _vp, _i32, _i32, C.c_uint64, _vp, _vp]),
    "svk_cube_gather": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "svk_cube_gather_cmvn": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "svk_cosine_scores": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "svk_roc_workspace_bytes": (C.c_size_t, [_i64]),
    "svk_roc_eer": (C.c_int, [_vp, _vp, _vp, _i64, _vp, C.c_size_t, C.POINTER(C.c_double)]),
    "svk_l2_dist": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "svk_c3d2_stage1_lds_bytes": (C.c_size_t, []),
    "svk_c3d2_stage1": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "svk_c3d2_stage2": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "svk_c3d2_conv31": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "svk_c3d2_conv32t": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "svk_c3d2_conv41": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "svk_c3d2_conv42": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "svk_c3d2_fc5_workspace_floats": (C.c_size_t, [_i32]),
    "svk_c3d2_fc5": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "svk_comm_unique_id": (C.c_int, [_vp, C.c_char_p]),
    "svk_comm_init": (C.c_int, [_vp, C.c_char_p, _i32, _i32]),
    "svk_allgather_f32": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "svk_comm_destroy": (C.c_int, [_vp]),
    "svk_comm_info": (C.c_int, [_vp, C.POINTER(_i32)]),
    "svk_ingest_resample": (C.c_int, [_vp, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _i32,
                                      _i64, _i32, _vp]),
}

_lib = None


def load():
    """The loaded library (cached).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `make -C speaker_verification_amd/csrc`); there is no CPU fallback")
        # torch first, when it is there: its wheel carries its own HIP runtime, and a process must not end up with two
        # (libsvk.so loaded before torch would pull in /opt/rocm's copy; the second runtime then sees no device)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)            # AttributeError if the .so is stale
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.svk_version() != VERSION:
            raise RuntimeError(f"libsvk.so version {lib.svk_version()} does not match this package ({VERSION}): "
                               "rebuild it (make -C speaker_verification_amd/csrc)")
        _lib = lib
    return _lib


def check(rc, ctx=None):
    if rc != SVK_OK:
        msg = ""
        if ctx:
            raw = load().svk_last_error(ctx)
            msg = raw.decode("utf-8", "replace") if raw else ""
        raise SvkError(rc, msg)


def provenance():
    """What a measurement was made with: {"csrc_sha": sha256 over the kernel sources (csrc/*.hip, csrc/*.h, csrc/Makefile,
    include/svk.h: file names + bytes, sorted), "libsvk_sha": sha256 of the built library}, 16 hex digits each.  Profiles
    under profiles/ carry it (tools/summarize_prof.py) and bench.py marks a roofline row `stale` when the counters it uses
    were collected from other kernel sources than the ones it runs."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")) or f == "Makefile")
    files.append(os.path.join(os.path.dirname(here), "include", "svk.h"))
    h = hashlib.sha256()
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    out = {"csrc_sha": h.hexdigest()[:16], "libsvk_sha": None}
    if os.path.exists(LIB_PATH):
        with open(LIB_PATH, "rb") as fh:
            out["libsvk_sha"] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out
