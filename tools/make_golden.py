#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE in the build container.

Run from the repo root:  python tools/make_golden.py
Needs /root/reference (read-only); nothing from it is copied -- only inputs
(seeds / small arrays) and the reference's OUTPUTS are stored.  The GPU box has
no /root/reference, so tests read the committed .npz files.

Import recipe:
  * speechpy            : sys.path += /root/reference/speech_feature_extraction
  * model / siamese     : sys.path += /root/reference
  * evaluation / utils  : import `librosa`, `torchvision`, `webrtcvad`, which are
                          absent here and not on the path exercised; they are
                          satisfied with inert placeholder modules so that the
                          `import` statements succeed.  No function of those
                          libraries is ever called.
  * cmvnw / derivative  : the reference calls `np.lib.pad`, removed in NumPy 2;
                          the harness sets `np.lib.pad = np.pad` (Q10/Q11).
"""
import io
import os
import re
import sys
import types
from unittest import mock

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, "speech_feature_extraction"))
sys.path.insert(0, REF)

if not hasattr(np.lib, "pad"):
    np.lib.pad = np.pad

from speechpy import feature as rf, processing as rp, functions as rfn   # the REFERENCE  # noqa: E402

from speaker_verification_amd import synth                                # noqa: E402
from speaker_verification_amd.model import perturb_inference_state        # noqa: E402


def versions():
    import scipy
    import sklearn
    return np.array([f"numpy {np.__version__}", f"scipy {scipy.__version__}",
                     f"sklearn {sklearn.__version__}", f"torch {torch.__version__}"])


def speechpy_fixture():
    g = {"versions": versions()}
    fs = 16000

    # ---- functions.py ----
    hz = np.array([0.0, 120.5, 300.0, 1000.0, 4000.0, 7999.0, 8000.0])
    g["fn_hz"] = hz
    g["fn_mel"] = rfn.frequency_to_mel(hz)
    g["fn_hz_back"] = rfn.mel_to_frequency(g["fn_mel"])
    tri_x = np.linspace(3, 17, 29)
    g["fn_tri_x"] = tri_x
    g["fn_tri"] = rfn.triangle(tri_x, left=5, middle=9, right=15)
    zh = np.array([0.0, 1.5, -0.0, 1e-300, -3.0])
    g["fn_zh_in"] = zh
    g["fn_zh"] = rfn.zero_handling(zh)

    # ---- filterbanks ----
    g["fb_A"] = rf.filterbanks(40, 257, 16000, 0, 8000)           # config A (Q1: 0 -> 300)
    g["fb_B"] = rf.filterbanks(40, 513, 16000, 0, 8000)           # config B
    g["fb_C"] = rf.filterbanks(26, 257, 16000, 100.0, 7000.0)
    g["fb_D"] = rf.filterbanks(20, 129, 8000, None, None)

    # ---- short clip, stage by stage ----
    short = synth.noise_clip(11, 4000)
    g["short_seed"] = np.array([11, 4000])
    g["pre_short_i16"] = rp.preemphasis(short, shift=1, cof=0.98)
    g["pre_short_f32"] = rp.preemphasis((short / 32768.0).astype(np.float32), shift=1, cof=0.98)
    g["pre_short_shift3"] = rp.preemphasis(short, shift=3, cof=0.5)
    g["frames_nopad"] = rp.stack_frames(short.astype(float), fs, 0.020, 0.010,
                                        filter=lambda x: np.ones((x,)), zero_padding=False)
    g["frames_pad"] = rp.stack_frames(short.astype(float), fs, 0.020, 0.020,
                                      filter=lambda x: np.ones((x,)), zero_padding=True)
    g["frames_hamming"] = rp.stack_frames(short.astype(float), fs, 0.025, 0.010,
                                          filter=np.hamming, zero_padding=True)
    fr = g["frames_nopad"]
    g["fftmag_512"] = rp.fft_spectrum(fr, 512)
    g["pow_512"] = rp.power_spectrum(fr, 512)
    g["pow_1024"] = rp.power_spectrum(g["frames_hamming"], 1024)
    g["pow_256_crop"] = rp.power_spectrum(fr, 256)                  # flen 320 > nfft: rfft crops
    g["logpow_512_norm"] = rp.log_power_spectrum(fr, 512, normalize=True)
    g["logpow_512_raw"] = rp.log_power_spectrum(fr, 512, normalize=False)

    # ---- 1 s clips: mfe / lmfe / mfcc variants ----
    one = synth.noise_clip(12, 16000)
    g["one_seed"] = np.array([12, 16000])
    f, e = rf.mfe(one, fs)
    g["mfe_A_feat"], g["mfe_A_energy"] = f, e
    g["lmfe_A"] = rf.lmfe(one, fs)
    g["mfcc_A"] = rf.mfcc(one, fs)
    g["mfcc_A_nodc"] = rf.mfcc(one, fs, dc_elimination=False)
    g["mfcc_A_40"] = rf.mfcc(one, fs, num_cepstral=40)
    g["mfcc_A_pre"] = rf.mfcc(rp.preemphasis(one, cof=0.98), fs)
    g["mfcc_A_lowhigh"] = rf.mfcc(one, fs, num_filters=26, low_frequency=100.0, high_frequency=7000.0)
    one_f32 = (one / 32768.0).astype(np.float32)
    g["lmfe_B_f32"] = rf.lmfe(one_f32, fs, 0.025, 0.01, 40, 1024)
    g["mfcc_B_f32"] = rf.mfcc(one_f32, fs, 0.025, 0.01, 13, 40, 1024)
    spk = synth.speaker_clip(3, 1, 16000)
    g["spk_seed"] = np.array([3, 1, 16000])
    g["mfcc_A_spk"] = rf.mfcc(spk, fs)
    g["lmfe_B_spk"] = rf.lmfe(spk, fs, 0.025, 0.01, 40, 1024)
    g["mfcc_A_zero"] = rf.mfcc(np.zeros(1600, dtype=np.int16), fs)
    g["mfcc_A_tooshort"] = rf.mfcc(np.zeros(320, dtype=np.int16), fs)   # -> empty (0, 13)

    # ---- headline 3 s known answers ----
    rng = np.random.default_rng(0)
    sig = (rng.standard_normal(48000) * 3000).astype(np.int16)
    m = rf.mfcc(sig, fs)
    g["kat_mfcc_A_3s"] = m
    g["kat_mfcc_A_3s_pre_cmvn"] = rp.cmvn(rf.mfcc(rp.preemphasis(sig, cof=0.98), fs), True)
    g["kat_lmfe_B_3s"] = rf.lmfe(sig.astype(np.float32) / 32768, fs, 0.025, 0.01, 40, 1024)

    # ---- post-processing ----
    base = g["mfcc_A"]
    g["cmvn_mean"] = rp.cmvn(base, variance_normalization=False)
    g["cmvn_var"] = rp.cmvn(base, variance_normalization=True)
    wide = np.random.default_rng(5).random((50, 100))
    g["cmvn_wide_in"] = wide
    g["cmvn_wide_var"] = rp.cmvn(wide, variance_normalization=True)
    g["cmvnw_mean"] = rp.cmvnw(base, win_size=301, variance_normalization=False)
    g["cmvnw_var"] = rp.cmvnw(base, win_size=301, variance_normalization=True)
    g["cmvnw_var_w31"] = rp.cmvnw(base, win_size=31, variance_normalization=True)
    g["deriv_w2"] = rp.derivative_extraction(base, DeltaWindows=2)
    g["deriv_w3"] = rp.derivative_extraction(base, DeltaWindows=3)
    g["deriv_cube"] = rf.extract_derivative_feature(g["lmfe_A"])
    np.savez_compressed(os.path.join(OUT, "speechpy.npz"), **g)
    print("speechpy.npz", sum(v.nbytes for v in g.values()) // 1024, "KiB raw")


def _placeholder(name):
    mod = types.ModuleType(name)
    mod.__getattr__ = lambda attr: mock.MagicMock(name=f"{name}.{attr}")   # inert
    return mod


def _import_reference_app_modules():
    for name in ("librosa", "torchvision", "torchvision.transforms", "webrtcvad"):
        if name not in sys.modules:
            sys.modules[name] = _placeholder(name)
    import evaluation as ref_eval          # noqa  (imports utils, model)
    import siamese as ref_siamese          # noqa
    import utils as ref_utils              # noqa
    import vad as ref_vad                  # noqa
    import model as ref_model              # noqa
    return ref_eval, ref_siamese, ref_utils, ref_vad, ref_model


class _EnergyVad:
    """Object with webrtcvad's `is_speech(bytes, sample_rate)` signature that
    applies THIS BUILD's integer energy rule (oracle/vad_ref.py docstring)."""

    def __init__(self, threshold):
        self.threshold = int(threshold)

    def is_speech(self, frame_bytes, sample_rate):
        x = np.frombuffer(frame_bytes, dtype=np.int16).astype(np.int64)
        return bool(int(np.sum(x * x)) > self.threshold * x.shape[0])


def vad_fixture(ref_vad):
    g = {"versions": versions()}
    thr = 250000
    g["threshold"] = np.array([thr])
    cases = []
    clips = [("spk_0_0", synth.speaker_clip(0, 0)), ("spk_1_4", synth.speaker_clip(1, 4)),
             ("spk_7_2", synth.speaker_clip(7, 2)), ("spk_5_0_long", synth.speaker_clip(5, 0, 112000)),
             ("noise_loud", synth.noise_clip(3, 48000, 3000.0)),
             ("noise_quiet", synth.noise_clip(4, 48000, 100.0)),
             ("len_47999", synth.speaker_clip(2, 1, 47999)), ("len_48001", synth.speaker_clip(2, 2, 48001)),
             ("len_480", synth.noise_clip(6, 480)), ("len_481", synth.noise_clip(6, 481)),
             ("len_100", synth.noise_clip(6, 100))]
    # a hand-made flag pattern that exercises trigger, release and re-trigger
    pat = np.array([0] * 3 + [1] * 9 + [0] + [1] * 12 + [0] * 9 + [1] + [0] * 11 + [1] * 10 + [0] * 4 + [1] * 15,
                   dtype=np.int64)
    loud = np.repeat(pat, 480) * 4000
    clips.append(("pattern", (loud * np.where(np.arange(loud.size) % 2, 1, -1)).astype(np.int16)))
    for name, pcm in clips:
        audio = pcm.tobytes()
        frames = list(ref_vad.frame_generator(30, audio, 16000))
        flags = np.array([_EnergyVad(thr).is_speech(fr.bytes, 16000) for fr in frames], dtype=bool)
        sink = io.StringIO()
        real_stdout, sys.stdout = sys.stdout, sink                  # vad_collector prints per frame
        try:
            segments = list(ref_vad.vad_collector(16000, 30, 300, _EnergyVad(thr), frames))
        finally:
            sys.stdout = real_stdout
        g[name + "_nframes"] = np.array([len(frames)])
        g[name + "_flags"] = flags
        g[name + "_seglens"] = np.array([len(s) // 2 for s in segments], dtype=np.int64)
        voiced = np.frombuffer(b"".join(segments), dtype=np.int16)
        g[name + "_voiced_sum"] = np.array([voiced.astype(np.int64).sum(), voiced.size,
                                            (voiced.astype(np.int64) ** 2).sum()])
        # keep mask from the collector's own trace: it prints one 0/1 per frame,
        # '+(t)' with the timestamp of the first frame of a segment when it
        # triggers and '-(t)' with the END time of the last frame when it
        # releases (vad.py:92,101,119,123); timestamps advance by exactly 0.03 s.
        keep = np.zeros(len(frames), dtype=bool)
        seg = np.full(len(frames), -1, dtype=np.int32)
        marks = re.findall(r"([+-])\(([0-9.eE+-]+)\)", sink.getvalue())
        assert len(marks) % 2 == 0, (name, marks)
        for k in range(0, len(marks), 2):
            assert marks[k][0] == "+" and marks[k + 1][0] == "-", (name, marks)
            lo = int(round(float(marks[k][1]) / 0.03))
            hi = int(round(float(marks[k + 1][1]) / 0.03))
            keep[lo:hi] = True
            seg[lo:hi] = k // 2
        assert int(keep.sum()) * 480 == len(b"".join(segments)) // 2, name
        assert [int((seg == k).sum()) * 480 for k in range(len(segments))] == \
            [len(s) // 2 for s in segments], name
        rebuilt = b"".join(fr.bytes for i, fr in enumerate(frames) if keep[i])
        assert rebuilt == b"".join(segments), name
        g[name + "_keep"] = keep
        g[name + "_seg"] = seg
        if name.startswith("len_") or name.startswith("spk") or name.startswith("noise") or name == "pattern":
            g[name + "_pcm_len"] = np.array([pcm.size])
        cases.append(name)
    g["pattern_pcm"] = clips[-1][1]
    g["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "vad.npz"), **g)
    print("vad.npz", sum(v.nbytes for v in g.values()) // 1024, "KiB raw")


def model_fixture(ref_model, ref_utils):
    g = {"versions": versions()}
    sink = io.StringIO()
    real_stdout, sys.stdout = sys.stdout, sink                      # C3D2.__init__ prints
    try:
        torch.manual_seed(2024)
        net = ref_model.C3D2(1211, 1)
    finally:
        sys.stdout = real_stdout
    state = perturb_inference_state(net.state_dict(), seed=99)
    net.load_state_dict(state)
    net.eval()
    g["init_seed"] = np.array([2024])
    g["perturb_seed"] = np.array([99])
    g["n_labels"] = np.array([1211])
    names = sorted(state.keys())
    g["state_names"] = np.array(names)
    g["state_abs_sums"] = np.array([float(state[k].double().abs().sum()) for k in names])
    cube_rng = np.random.default_rng(31)
    cubes = (cube_rng.standard_normal((3, 1, 20, 80, 40)) * 2.0 - 6.0).astype(np.float32)
    g["cube_seed"] = np.array([31])
    with torch.no_grad():
        g["embed"] = net(torch.from_numpy(cubes), development=False).numpy()
        g["softmax_row0_top"] = net(torch.from_numpy(cubes[:1]), development=True).numpy()[0, :8]
        g["speaker_model"] = net.create_Speaker_Model(torch.from_numpy(cubes[1:2])).numpy()

    # FeatureCube with the reference's RNG protocol (utils.py:15,372)
    feat = np.random.default_rng(32).standard_normal((297, 40))
    np.random.seed(777)
    sample = ref_utils.FeatureCube((80, 40, 20))({"feature": feat, "label": 5})
    g["cube_feat_seed"] = np.array([32])
    g["cube_np_seed"] = np.array([777])
    g["cube_out"] = sample["feature"]
    np.random.seed(777)
    g["cube_idx"] = np.random.randint(297 - 80, size=20)
    np.savez_compressed(os.path.join(OUT, "c3d2_embed.npz"), **g)
    print("c3d2_embed.npz", sum(v.nbytes for v in g.values()) // 1024, "KiB raw")


def scoring_fixture(ref_eval, ref_siamese):
    g = {"versions": versions()}
    rng = np.random.default_rng(41)
    n_spk, per = 6, 7
    centres = rng.standard_normal((n_spk, 128)).astype(np.float32)
    test = np.repeat(centres, per, axis=0) + 1.5 * rng.standard_normal((n_spk * per, 128)).astype(np.float32)
    test = test.astype(np.float32)
    enroll = (centres + 0.8 * rng.standard_normal((n_spk, 128))).astype(np.float32)
    g["test"], g["enroll"] = test, enroll

    class _Fixed(torch.nn.Module):                                   # stands for the embedding net
        def forward(self, utterance, development=False):
            return utterance

    ev = object.__new__(ref_eval.Evaluation)                         # __init__ reads .pt files from disk
    ev.model = _Fixed()
    ev.speaker_models = {f"id{j:05d}": torch.from_numpy(enroll[j:j + 1]) for j in range(n_spk)}
    sims = np.zeros((test.shape[0], n_spk))
    assigned = np.zeros((test.shape[0], n_spk))
    for i in range(test.shape[0]):
        sims[i], assigned[i] = ev.compute_Similarity(torch.from_numpy(test[i:i + 1]))
    g["sims"], g["assigned"] = sims, assigned
    labels = np.zeros_like(sims)
    labels[np.arange(test.shape[0]), np.repeat(np.arange(n_spk), per)] = 1
    g["labels"] = labels
    eer, auc, fpr, tpr = ref_eval.get_eer_auc(labels.flatten(), sims.flatten())
    g["eer"], g["auc"], g["fpr"], g["tpr"] = np.array([eer]), np.array([auc]), fpr, tpr
    # a second, larger, noisier problem for EER only
    big_s = rng.standard_normal(4000) + np.repeat([0.0, 1.2], 2000)
    big_l = np.repeat([0.0, 1.0], 2000)
    eer2, auc2, _, _ = ref_eval.get_eer_auc(big_l, big_s)
    g["big_scores"], g["big_labels"] = big_s, big_l
    g["big_eer"], g["big_auc"] = np.array([eer2]), np.array([auc2])

    sia = ref_siamese.Siamese(LAMBDA=0.001, M=2.0)
    o1 = rng.standard_normal((9, 128)).astype(np.float32)
    o2 = rng.standard_normal((9, 128)).astype(np.float32)
    g["l2_o1"], g["l2_o2"] = o1, o2
    g["l2_dist"] = sia.l2_dist(torch.from_numpy(o1), torch.from_numpy(o2)).numpy()

    # ---- Siamese.forward (siamese.py:10-27) as written.  Its only CUDA dependence is the `.cuda()` ATTRIBUTE
    # (siamese.py:16,21): with a harness-side `torch.Tensor.cuda = identity` (same category as `np.lib.pad = np.pad`
    # above; the reference is not edited) the reference's own forward runs on the CPU, prints its two lines and
    # returns the loss.  The model only contributes its parameter norms: the reference's C3D2 under a fixed seed. ----
    sink = io.StringIO()
    real_stdout, sys.stdout = sys.stdout, sink
    try:
        torch.manual_seed(77)
        import model as ref_model
        net = ref_model.C3D2(100, 1)
    finally:
        sys.stdout = real_stdout
    g["sf_model_seed"] = np.array([77, 100, 1])
    g["sf_param_norms"] = np.array([float(torch.norm(p_)) for p_ in net.parameters()], dtype=np.float64)
    n_pairs = 12
    e1 = (0.11 * rng.standard_normal((n_pairs, 128))).astype(np.float32)
    e2 = (0.11 * rng.standard_normal((n_pairs, 128))).astype(np.float32)
    e2[:3] = e1[:3] + (0.01 * rng.standard_normal((3, 128))).astype(np.float32)     # three near-identical pairs
    yy = np.array([1, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1], dtype=np.float32)
    g["sf_o1"], g["sf_o2"], g["sf_y"] = e1, e2, yy
    cases = [(0.001, 2.0), (0.01, 1.0), (0.0, 1.75), (0.05, 0.05)]    # M = 1.0 / 0.05: most / all impostor pairs have d > M
    g["sf_cases"] = np.array(cases, dtype=np.float64)
    saved_cuda = getattr(torch.Tensor, "cuda")
    torch.Tensor.cuda = lambda self, *a, **k: self
    losses, same, notsame = [], [], []
    try:
        for lam, m in cases:
            sink = io.StringIO()
            real_stdout, sys.stdout = sys.stdout, sink
            try:
                with torch.no_grad():
                    loss = ref_siamese.Siamese(LAMBDA=lam, M=m).forward(net, torch.from_numpy(yy), torch.from_numpy(e1),
                                                                         torch.from_numpy(e2))
            finally:
                sys.stdout = real_stdout
            losses.append(float(loss))
            same.append(float(re.search(r"^Same: (\S+)", sink.getvalue(), re.M).group(1)))
            notsame.append(float(re.search(r"^Not same: (\S+)", sink.getvalue(), re.M).group(1)))
    finally:
        torch.Tensor.cuda = saved_cuda
    g["sf_loss"], g["sf_same_mean"], g["sf_notsame_mean"] = np.array(losses), np.array(same), np.array(notsame)
    g["sf_dist"] = sia.l2_dist(torch.from_numpy(e1), torch.from_numpy(e2)).numpy()
    np.savez_compressed(os.path.join(OUT, "scoring.npz"), **g)
    print("scoring.npz", sum(v.nbytes for v in g.values()) // 1024, "KiB raw")


def _collector_masks(ref_vad, pcm, frame_ms, padding_ms, thr):
    """keep / segment masks of the REFERENCE's vad_collector (vad.py:60-129) for any frame / padding
    duration, read off its own trace ('+(t)' start time of a segment, '-(t)' end time)."""
    audio = pcm.tobytes()
    frames = list(ref_vad.frame_generator(frame_ms, audio, 16000))
    sink = io.StringIO()
    real_stdout, sys.stdout = sys.stdout, sink
    try:
        segments = list(ref_vad.vad_collector(16000, frame_ms, padding_ms, _EnergyVad(thr), frames))
    finally:
        sys.stdout = real_stdout
    dur = frame_ms / 1000.0
    n = len(frames[0].bytes) // 2 if frames else 0
    keep = np.zeros(len(frames), dtype=bool)
    seg = np.full(len(frames), -1, dtype=np.int32)
    marks = re.findall(r"([+-])\(([0-9.eE+-]+)\)", sink.getvalue())
    for k in range(0, len(marks) - 1, 2):
        assert marks[k][0] == "+" and marks[k + 1][0] == "-", marks
        lo, hi = int(round(float(marks[k][1]) / dur)), int(round(float(marks[k + 1][1]) / dur))
        keep[lo:hi] = True
        seg[lo:hi] = k // 2
    if len(marks) % 2:                      # a segment still open at the end: no '-(t)' is printed (vad.py:126-129)
        assert marks[-1][0] == "+"
        lo = int(round(float(marks[-1][1]) / dur))
        keep[lo:] = True
        seg[lo:] = len(marks) // 2
    rebuilt = b"".join(fr.bytes for i, fr in enumerate(frames) if keep[i])
    assert rebuilt == b"".join(segments), "trace parsing disagrees with the yielded segments"
    assert [int((seg == k).sum()) * n for k in range(len(segments))] == [len(s) // 2 for s in segments]
    return keep, seg


class _Compose:
    """what `torchvision.transforms.Compose` does (torchvision is absent here)"""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample


def _wav_as_librosa(path, sample_rate=16000):
    """What `librosa.load(path, sr=16000, mono=True)[0]` returns for a 16 kHz mono 16-bit WAV: the samples
    as float32 / 32768, no resampling (librosa is absent here; files of this fixture are all of that kind)."""
    import wave
    with wave.open(path, "rb") as wf:
        assert wf.getframerate() == sample_rate and wf.getnchannels() == 1 and wf.getsampwidth() == 2
        pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)
    return pcm.astype(np.float32) / np.float32(32768.0)


def round2_fixture(ref_vad, ref_eval, ref_model, ref_utils):
    """Round-2 additions: nfft 1024 at other sampling rates (bank reaches bin 256), a VAD ring longer than
    64 frames, and the reference's FILE-DRIVEN create_speaker_models() / evaluate() on a synthetic tree."""
    import tempfile
    g = {"versions": versions()}
    # ---- fft_length 1024 away from 16 kHz (feature.py:77-99: the bank's last filter ends on bin 256) ----
    for fs in (8000, 32000, 44100):
        sig = synth.speaker_clip(9, fs // 1000, fs // 2, fs)          # 0.5 s at that rate
        g[f"lmfe_1024_fs{fs}"] = rf.lmfe(sig, fs, 0.025, 0.01, 40, 1024)
        g[f"mfcc_1024_fs{fs}"] = rf.mfcc(sig, fs, fft_length=1024)
        f, e = rf.mfe((sig / 32768.0).astype(np.float32), fs, fft_length=1024, num_filters=26)
        g[f"mfe_1024_fs{fs}_feat"], g[f"mfe_1024_fs{fs}_energy"] = f, e
    # ---- 10 ms frames with 1 s of padding: a ring of 100 frames ----
    thr = 250000
    g["vad_threshold"] = np.array([thr])
    pat = np.array([0] * 30 + [1] * 120 + [0] * 130 + [1] * 95 + [0] * 20 + [1] * 200 + [0] * 150 + [1] * 50 + [0] * 7,
                   dtype=np.int64)
    loud = np.repeat(pat, 160) * 4000
    g["vad_pattern10_pcm"] = (loud * np.where(np.arange(loud.size) % 2, 1, -1)).astype(np.int16)
    for name, pcm in (("spk_0_0", synth.speaker_clip(0, 0)), ("spk_5_0_long", synth.speaker_clip(5, 0, 112000)),
                      ("noise_loud", synth.noise_clip(3, 48000, 3000.0)), ("spk_3_1", synth.speaker_clip(3, 1, 80000)),
                      ("pattern10", g["vad_pattern10_pcm"])):
        for frame_ms, pad_ms in ((10, 1000), (10, 700), (20, 1500)):
            keep, seg = _collector_masks(ref_vad, pcm, frame_ms, pad_ms, thr)
            g[f"vad_{name}_{frame_ms}_{pad_ms}_keep"] = keep
            g[f"vad_{name}_{frame_ms}_{pad_ms}_seg"] = seg
    # ---- file-driven enrolment + evaluation (model.py:351-388, evaluation.py:90-146) ----
    import constants as ref_c
    import load_data as ref_load_data
    with tempfile.TemporaryDirectory() as root:
        data_dir, rel, state = synth.write_verification_tree(root)
        ref_c.ROOT, ref_c.DATA_ORIGIN = root, data_dir
        ref_load_data.load_wav = _wav_as_librosa                         # librosa.load stand-in, see its docstring
        ref_utils.transforms = types.SimpleNamespace(Compose=_Compose)   # torchvision stand-in
        captured = {}
        real_plot = ref_eval.get_and_plot_k_eer_auc

        def spy(label, scores, k=1):
            captured["labels"], captured["scores"] = np.array(label), np.array(scores)
            return real_plot(label, scores, k)

        cwd = os.getcwd()
        sink = io.StringIO()
        real_stdout, sys.stdout = sys.stdout, sink
        try:
            os.chdir(root)                                               # evaluate() saves eer_auc.png in the CWD
            np.random.seed(4242)
            ref_model.create_speaker_models()
            order = [f.replace(".pt", "") for f in os.listdir(os.path.join(root, "speaker_models"))]
            enrolled = np.concatenate([torch.load(os.path.join(root, "speaker_models", sid + ".pt")).detach().numpy()
                                       for sid in order])
            np.random.seed(4343)
            ref_eval.get_and_plot_k_eer_auc = spy
            ref_eval.evaluate()
        finally:
            ref_eval.get_and_plot_k_eer_auc = real_plot
            sys.stdout = real_stdout
            os.chdir(cwd)
        out = sink.getvalue()
        g["eval_speaker_order"] = np.array(order)
        g["eval_enrolled"] = enrolled
        n_spk = len(order)
        g["eval_scores"] = captured["scores"].reshape(-1, n_spk)
        g["eval_labels"] = captured["labels"].reshape(-1, n_spk)
        g["eval_eer_pct"] = np.array([float(re.search(r"EER= ([0-9.eE+-]+)", out).group(1))])
        g["eval_auc_pct"] = np.array([float(re.search(r"AUC= ([0-9.eE+-]+)", out).group(1))])
        g["eval_accuracy_pct"] = np.array([float(re.search(r"Accuracy: ([0-9.eE+-]+)%", out).group(1))])
        g["eval_closest"] = np.array(re.findall(r"the speaker was closer to (\S+)", out))
        g["eval_seeds"] = np.array([4242, 4343])
    np.savez_compressed(os.path.join(OUT, "round2.npz"), **g)
    print("round2.npz", sum(v.nbytes for v in g.values()) // 1024, "KiB raw")


def round4_fixture(ref_model, ref_utils):
    """The committed TRAINED checkpoint (speaker_verification_amd/checkpoints/c3d2_synth.pt, written by this build's
    tools/train_synth_checkpoint.py; loaded weights-only) through the REFERENCE's own `C3D2(100, 1).load_checkpoint(...)`
    (model.py:177-186) and forward, on four synthetic clips run through the reference's speechpy chain as the checkpoint
    was trained: preemphasis(x / 32768, cof=0.98) -> lmfe(16000, 0.025, 0.01, 40, 1024) -> cmvn(variance) ->
    utils.FeatureCube((80, 40, 20)) under a seeded global NumPy RNG."""
    import hashlib
    g = {"versions": versions()}
    path = os.path.join(REPO, "speaker_verification_amd", "checkpoints", "c3d2_synth.pt")
    g["checkpoint_sha256"] = np.array([hashlib.sha256(open(path, "rb").read()).hexdigest()])
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sink = io.StringIO()
    real_stdout, sys.stdout = sys.stdout, sink                      # C3D2.__init__ prints
    try:
        net = ref_model.C3D2(100, 1).load_checkpoint(ck)
    finally:
        sys.stdout = real_stdout
    net.eval()
    clips = [(3, 0), (3, 1), (4, 0), (4, 1)]                         # (synth speaker, utterance): synth.speaker_clip
    g["clip_ids"] = np.array(clips)
    g["np_seed"] = np.array([4242])
    np.random.seed(4242)
    cubes, idxs, sums = [], [], []
    for spk, utt in clips:
        sig = synth.speaker_clip(spk, utt) / 32768.0
        feat = rf.lmfe(rp.preemphasis(sig, cof=0.98), sampling_frequency=16000, frame_length=0.025, frame_stride=0.01,
                       num_filters=40, fft_length=1024)
        feat = rp.cmvn(feat, variance_normalization=True)
        state = np.random.get_state()
        idxs.append(np.random.randint(feat.shape[0] - 80, size=20))   # the draw FeatureCube is about to make (utils.py:372)
        np.random.set_state(state)
        cubes.append(ref_utils.FeatureCube((80, 40, 20))({"feature": feat, "label": 0})["feature"])
        sums.append([feat.shape[0], float(feat.sum()), float(np.abs(feat).sum())])
    cubes = np.stack(cubes)                                          # (4, 1, 20, 80, 40) float32
    g["crop_idx"] = np.stack(idxs).astype(np.int32)
    g["feat_frames_sum_abssum"] = np.array(sums)
    g["cube_abssum"] = np.array([float(np.abs(cubes[k]).sum()) for k in range(4)])
    with torch.no_grad():
        g["embed"] = net(torch.from_numpy(cubes), development=False).numpy()
        g["softmax_top"] = net(torch.from_numpy(cubes), development=True).numpy()[:, :8]
        g["speaker_model"] = net.create_Speaker_Model(torch.from_numpy(cubes[3:4])).numpy()
    from sklearn.metrics.pairwise import cosine_similarity            # what evaluation.py:77 calls
    g["cosine"] = cosine_similarity(g["embed"], g["embed"])
    np.savez_compressed(os.path.join(OUT, "round4.npz"), **g)
    print("round4.npz", sum(v.nbytes for v in g.values()) // 1024, "KiB raw; same / different speaker cosine:",
          float(g["cosine"][0, 1]), float(g["cosine"][0, 2]))


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]                      # e.g. `python tools/make_golden.py round2`
    ref_eval, ref_siamese, ref_utils, ref_vad, ref_model = _import_reference_app_modules()
    if not only or "speechpy" in only:
        speechpy_fixture()
    if not only or "vad" in only:
        vad_fixture(ref_vad)
    if not only or "model" in only:
        model_fixture(ref_model, ref_utils)
    if not only or "scoring" in only:
        scoring_fixture(ref_eval, ref_siamese)
    if not only or "round2" in only:
        round2_fixture(ref_vad, ref_eval, ref_model, ref_utils)
    if not only or "round4" in only:
        round4_fixture(ref_model, ref_utils)


if __name__ == "__main__":
    main()
