"""Host-side logic of the drop-in layer that needs no GPU: table builders, frame
bookkeeping, sharding arithmetic, the synthetic corpus, model plumbing."""
import os

import numpy as np
import pytest
import torch

from oracle import speechpy_ref as ref
from speaker_verification_amd import distributed as svdist, synth
from speaker_verification_amd._lib import OUT_LMFE, OUT_MFCC
from speaker_verification_amd.engine import spec_from_seconds
from speaker_verification_amd.speechpy import feature, functions


def test_filterbank_tables_match_reference(golden):
    g = golden["speechpy"]
    np.testing.assert_array_equal(feature.filterbanks(40, 257, 16000, 0, 8000), g["fb_A"])
    np.testing.assert_array_equal(feature.filterbanks(40, 513, 16000, 0, 8000), g["fb_B"])
    np.testing.assert_array_equal(feature.filterbanks(26, 257, 16000, 100.0, 7000.0), g["fb_C"])
    np.testing.assert_array_equal(feature.filterbanks(20, 129, 8000, None, None), g["fb_D"])
    with pytest.raises(AssertionError):
        feature.filterbanks(40, 257, 16000, 0, 9000)
    np.testing.assert_array_equal(functions.frequency_to_mel(g["fn_hz"]), g["fn_mel"])
    np.testing.assert_array_equal(functions.mel_to_frequency(g["fn_mel"]), g["fn_hz_back"])
    np.testing.assert_array_equal(functions.triangle(g["fn_tri_x"], 5, 9, 15), g["fn_tri"])
    np.testing.assert_array_equal(functions.zero_handling(g["fn_zh_in"]), g["fn_zh"])


def test_frontend_spec_frame_counts():
    a = spec_from_seconds(16000, 0.020, 0.01, 512, 40, 13, OUT_MFCC)
    b = spec_from_seconds(16000, 0.025, 0.01, 1024, 40, 40, OUT_LMFE)
    assert (a.frame_len, a.frame_stride, a.num_frames(48000), a.num_cols) == (320, 160, 298, 13)
    assert (b.frame_len, b.frame_stride, b.num_frames(48000), b.num_cols) == (400, 160, 297, 40)
    for n in (0, 100, 319, 320, 479, 480, 481, 16000, 47999, 48001):
        want = max(0, ref.frame_geometry(n, 16000, 0.020, 0.01, False)[2])
        assert a.num_frames(n) == want, n
    assert a.key() == spec_from_seconds(16000, 0.020, 0.01, 512, 40, 13, OUT_MFCC).key()
    assert a.key() != b.key()


def test_shard_bounds_cover_everything():
    for n, w in ((148642, 8), (148642, 1), (4874, 4), (7, 8), (0, 2), (16, 2)):
        spans = [svdist.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) <= svdist.shard_rows(n, w)
    assert svdist.shard_rows(148642, 8) == 18581
    assert svdist.shard_bounds(148642, 8, 7) == (130067, 148642)


def test_synth_is_deterministic_and_vad_friendly():
    from oracle import vad_ref
    a, b = synth.speaker_clip(3, 2), synth.speaker_clip(3, 2)
    assert a.dtype == np.int16 and a.shape == (48000,) and np.array_equal(a, b)
    assert not np.array_equal(a, synth.speaker_clip(3, 3))
    pcm, spk = synth.corpus(3, 2)
    assert pcm.shape == (6, 48000) and list(spk) == [0, 0, 1, 1, 2, 2]
    for s in range(12):
        keep, _, voiced = vad_ref.vad_energy(synth.speaker_clip(s, 0))
        assert keep.any(), s
        assert ref.frame_geometry(voiced.size, 16000, 0.025, 0.01, False)[2] > 80, s   # enough frames for the cube


def test_model_matches_reference_layout():
    from speaker_verification_amd.model import C3D2, seeded_model
    m = seeded_model(1, n_labels=10)
    assert sum(p.numel() for p in C3D2(100, 1).parameters()) == 1164413
    keys = list(m.state_dict().keys())
    assert keys[0] == "conv1_1.weight" and "batch_norm4_2.running_var" in keys and "FC6.bias" in keys
    x = torch.randn(2, 1, 20, 80, 40)
    with torch.no_grad():
        assert m(x, development=False).shape == (2, 128)
        probs = m(x)
        assert probs.shape == (2, 10) and torch.allclose(probs.sum(1), torch.ones(2), atol=1e-5)
    # checkpoint format of the reference: {'state_dict': ...} with DataParallel prefixes (model.py:177-186)
    ckpt = {"state_dict": {"module." + k: v for k, v in m.state_dict().items()}}
    m2 = C3D2(10, 1).load_checkpoint(ckpt).cpu().eval()
    with torch.no_grad():
        assert torch.equal(m2(x, development=False), m(x, development=False))


def test_enroll_last_utterance():
    from speaker_verification_amd.pipeline import enroll_last_utterance
    ids, last = enroll_last_utterance(None, np.array([5, 5, 2, 5, 2, 9]))
    assert list(ids) == [2, 5, 9] and list(last) == [4, 3, 5]                    # Q17: last one wins


def test_fused_embedder_folds_batchnorm():
    """FusedEmbedder's BN-folded weights: conv(x, w', b') equals BatchNorm(conv(x, w, b)) (eval mode) for every layer -- the
    tables the libsvk kernels consume are built from these.  Runs on the host (the tables are plain tensor algebra); the
    kernels themselves need the GPU and say so."""
    import torch.nn.functional as F
    from speaker_verification_amd.model import perturb_inference_state, seeded_model
    m = seeded_model(3, n_labels=4)
    m.load_state_dict(perturb_inference_state(m.state_dict(), 4))
    fused = m.fused_inference()
    assert fused is m.fused_inference()                                   # cached while the weights do not change
    gen = torch.Generator().manual_seed(1)
    for li, tag in enumerate(("1_1", "1_2", "2_1", "2_2", "3_1", "3_2", "4_1", "4_2")):
        conv, bn = getattr(m, "conv" + tag), getattr(m, "batch_norm" + tag)
        w, b, slope, stride = fused.stages[li][:4]
        x = torch.randn((1, conv.in_channels, 5, 12, 9), generator=gen)
        with torch.no_grad():
            want = bn(conv(x))
            got = F.conv3d(x, w, b, stride=stride)
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
        assert torch.equal(slope, getattr(m, "PReLu" + tag).weight.detach())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fused(torch.zeros(1, 1, 20, 80, 40))
    with torch.no_grad():
        m.conv1_1.weight.mul_(1.5)                                        # an in-place update invalidates the snapshot
    assert m.fused_inference() is not fused


def test_ingest_host_tables_and_wav_reader(tmp_path):
    """Host side of the audio ingest: rate ratio, FIR taps (= the oracle's), any-rate / any-channel WAV reader."""
    import wave
    from oracle import ingest_ref
    from speaker_verification_amd import ingest
    assert ingest.rational_ratio(44100, 16000) == (160, 441)
    assert ingest.rational_ratio(48000, 16000) == (1, 3)
    assert ingest.rational_ratio(8000, 16000) == (2, 1)
    assert ingest.rational_ratio(16000, 16000) == (1, 1)
    with pytest.raises(ValueError):
        ingest.rational_ratio(0, 16000)
    for up, down in [(1, 3), (160, 441), (2, 1)]:
        np.testing.assert_allclose(ingest.resample_taps(up, down), ingest_ref.firwin_kaiser(up, down), rtol=0, atol=1e-15)
        assert len(ingest.resample_taps(up, down)) == 20 * max(up, down) + 1
    frames = (np.arange(2 * 500).reshape(500, 2) * 13 % 2000 - 1000).astype(np.int16)
    path = str(tmp_path / "stereo44k.wav")
    with wave.open(path, "wb") as wf:
        wf.setnchannels(2)
        wf.setsampwidth(2)
        wf.setframerate(44100)
        wf.writeframes(frames.tobytes())
    got, rate = ingest.read_wave_any(path)
    assert rate == 44100 and got.shape == (500, 2) and np.array_equal(got, frames)


def test_indexed_labels_never_unpickle(tmp_path):
    """The id table of the file-driven entry points: json if present, else derived from the id list; the reference's
    pickled .npy is never opened."""
    import json
    from speaker_verification_amd.evaluation import load_indexed_labels
    from speaker_verification_amd import synth
    root = str(tmp_path)
    synth.write_verification_tree(root, n_speakers=3, utts_per_speaker=2, n_samples=16000)
    want = {"id10001": 0, "id10002": 1, "id10003": 2}
    assert load_indexed_labels(root + "/50_first_ids.npy") == want
    os.remove(root + "/50_first_ids.json")
    with open(root + "/50_first_ids.npy", "wb") as fh:
        fh.write(b"not a numpy file: must never be opened")
    assert load_indexed_labels(root + "/50_first_ids.npy") == want
    os.remove(root + "/50_first_ids.txt")
    with pytest.raises(FileNotFoundError, match="pickle"):
        load_indexed_labels(root + "/50_first_ids.npy")


def test_ragged_batches_sort_cap_and_merge_the_tail():
    """`VerificationPipeline._ragged_batches` (host logic of embed_ragged / embed_ragged_resident): clips sorted by length,
    batches capped by clip count and by 16-byte-aligned samples, and a last batch of a handful of (long) clips joined to its
    predecessor instead of running the network on a few cubes."""
    import types
    from speaker_verification_amd.pipeline import VerificationPipeline
    fn = VerificationPipeline._ragged_batches
    me = types.SimpleNamespace(micro_batch=100)
    lens = [1000] * 270 + [50_000] * 3
    inf = float("inf")
    out = fn(me, lens, max_batch_samples=200_000, max_padding=inf)
    flat = [k for b, _ in out for k in b]
    assert sorted(flat) == list(range(len(lens)))                        # every clip exactly once
    assert [lens[k] for k in flat] == sorted(lens)                       # ascending length
    assert len(out[0][0]) == 100 and out[0][1] == 100 * 1000             # clip-count cap; 1000 is a multiple of 8
    # greedy: 100 + 100 + (70 short + 2 long = 170 000 samples) + (1 long); the last, a single clip, joins its predecessor
    assert [len(b) for b, _ in out] == [100, 100, 73]
    assert out[2][1] == 70 * 1000 + 3 * 50_000
    # with the padding cap: 70 short clips are not padded to the length of the long ones
    out = fn(me, lens, max_batch_samples=200_000, max_padding=1.6)
    assert [len(b) for b, _ in out] == [100, 100, 70, 3] and sorted(k for b, _ in out for k in b) == list(range(len(lens)))
    for b, total in out:
        assert len(b) * max(lens[k] for k in b) <= 1.6 * total
    # the sample cap; a short tail joins its predecessor up to 1.5 x the cap together, not beyond
    assert [len(b) for b, _ in fn(me, [60_000] * 5, max_batch_samples=130_000)] == [2, 3]
    assert [len(b) for b, _ in fn(me, [100_000] * 3, max_batch_samples=120_000)] == [1, 1, 1]
    out = fn(me, [7], max_batch_samples=10)                               # one short clip: one batch, slot rounded up to 8
    assert out == [([0], 8)]
    assert fn(me, [], 10) == []
    # a tail never takes a batch past `micro_batch` clips (the network step of the cube ring) ...
    assert [len(b) for b, _ in fn(me, [1000] * 103, max_batch_samples=10**9)] == [100, 3]
    # ... nor multiplies its feature buffer: 5 clips of 145 s joined to 90 clips of 20 s make it 95 x 14 500 frames (220 MB)
    lens = [320_000] * 90 + [2_320_000] * 5
    assert [len(b) for b, _ in fn(me, lens, max_batch_samples=30_000_000, max_feature_bytes=64 << 20, max_padding=inf)] == [90, 5]
    assert [len(b) for b, _ in fn(me, lens, max_batch_samples=30_000_000, max_padding=inf)] == [95]
    assert [len(b) for b, _ in fn(me, lens, max_batch_samples=30_000_000, max_padding=1.6)] == [90, 5]
    # a VoxCeleb-like length distribution: every clip once, ascending, every batch within the caps
    rng = np.random.default_rng(7)
    lens = (np.minimum(4.0 + rng.lognormal(np.log(3.2), 0.85, 3000), 145.0) * 16000).astype(np.int64)
    me = types.SimpleNamespace(micro_batch=1024)
    out = fn(me, lens, max_batch_samples=64 << 20, max_padding=2.0)
    flat = [k for b, _ in out for k in b]
    assert sorted(flat) == list(range(3000)) and all(lens[a] <= lens[b] for a, b in zip(flat, flat[1:]))
    for b, total in out:
        assert len(b) <= 1024 and total == int(((lens[b] + 7) // 8 * 8).sum()) and (total <= 64 << 20 or len(b) == 1)
        assert len(b) * ((int(lens[b].max()) + 7) // 8 * 8) <= 2.0 * total + 8 or len(b) < 64
    # small tails join even past the padding cap while their PADDED size stays within one batch's real size: the benchmark's
    # 2 048 lengths end in 13 clips of 31 .. 45 s + the one 145 s clip -- one batch of 14 (padded 2 030 s), not two, and not
    # joined to the 255 clips of 12 .. 30 s in front of them (that would pad 269 clips to 145 s)
    import bench
    lens = bench.ragged_lengths(2048)
    out = fn(me, lens, max_batch_samples=64 << 20)
    assert [len(b) for b, _ in out] == [769, 583, 427, 255, 14] and int(lens[out[-1][0]].max()) == 145 * 16000
    assert sorted(k for b, _ in out for k in b) == list(range(2048))


def test_tail_operand_tables_against_naive_indexing():
    """The host tables of svk_c3d2_conv42 / fc5 (model.FusedEmbedder): depth-transformed weights in the MFMA lane order of
    include/svk.h, FC5's columns permuted to conv4_2's chunked output order -- against element-wise indexing."""
    import random
    import torch
    from speaker_verification_amd.model import perturb_inference_state, seeded_model
    m = seeded_model(3, 8)
    m.load_state_dict(perturb_inference_state(m.state_dict(), 4))
    e = m.fused_inference()
    rnd = random.Random(1)
    for name, li, axis in (("conv42", 7, "h"),):
        frag = getattr(e, name + "_tables")()[0]
        w = e.stages[li][0]
        g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
        g = torch.stack((g0, 0.5 * ((g0 + g2) + g1), 0.5 * ((g0 + g2) - g1), g2))        # [k][co][ci][kh][kw]
        g = g[:, :, :, :, 0] if axis == "h" else g[:, :, :, 0, :]
        assert tuple(frag.shape) == (w.shape[0] // 16, w.shape[1] // 8, g.shape[3], 4, 64, 2)
        for _ in range(500):
            nt, ch, tap = rnd.randrange(frag.shape[0]), rnd.randrange(frag.shape[1]), rnd.randrange(frag.shape[2])
            k, lane, el = rnd.randrange(4), rnd.randrange(64), rnd.randrange(2)
            assert frag[nt, ch, tap, k, lane, el] == g[k, 16 * nt + (lane & 15), 8 * ch + 2 * (lane >> 4) + el, tap]
    frag, bias = e.fc5_tables()
    assert tuple(frag.shape) == (4, 8, 72, 64, 4) and torch.equal(bias, e.fc_b)
    for _ in range(500):
        d, nt, st, lane, el = rnd.randrange(4), rnd.randrange(8), rnd.randrange(72), rnd.randrange(64), rnd.randrange(4)
        K = 1152 * d + 16 * st + 4 * (lane >> 4) + el               # ((d * 16 + chunk) * 9 + pixel) * 8 + c8
        c8, pix, chunk = K % 8, (K // 8) % 9, (K // 72) % 16
        assert frag[d, nt, st, lane, el] == e.fc_w[16 * nt + (lane & 15), (8 * chunk + c8) * 36 + d * 9 + pix]


def test_half_pair_weight_blocks_against_naive_indexing():
    """The host tables of the kernels that multiply through two-piece f16 products (svk_c3d2_stage1 / stage2 / conv31 / conv32t /
    conv41; include/svk.h): H = f16(w), L = f16(w - H) in the lane order of v_mfma_f32_16x16x32_f16's A operand -- element by element
    against the BN-folded weights, and H + L back to the weight within 2^-21 (or the last bit of an f16 subnormal)."""
    import random
    import torch
    from speaker_verification_amd.model import perturb_inference_state, seeded_model
    m = seeded_model(5, 8)
    m.load_state_dict(perturb_inference_state(m.state_dict(), 6))
    e = m.fused_inference()
    rnd = random.Random(2)

    def check(blk, want):          # blk [2: H | L] halves, want: the f32 weight the pair stands for
        h, l = blk[0].float(), blk[1].float()
        assert h == float(torch.tensor(want).to(torch.float16))
        assert abs((h + l) - want) <= max(2.0 ** -21 * abs(want), 2.0 ** -25)

    w1blk, b1, _, w2blk, b2 = e.stage1_tables()[:5]
    w1, w2 = e.stages[0][0], e.stages[1][0]
    assert tuple(w1blk.shape) == (2, 64, 8) and tuple(w2blk.shape) == (14, 2, 64, 8) and w1blk.dtype == torch.float16
    pairs = [((p // 4, 2 * (p % 4)), (p // 4, 2 * (p % 4) + 1)) for p in range(12)] + [((0, 8), (1, 8)), ((2, 8), None)]
    for _ in range(400):
        lane, el = rnd.randrange(64), rnd.randrange(8)
        co, kk = lane & 15, lane >> 4
        t = 8 * (kk & 1) + el                                        # conv1_1: tap t = 5 kd + kw, 15 = the zero column
        want = float(w1[co, 0, t // 5, 0, t % 5]) if t < 15 else 0.0
        assert float(w1blk[0, lane, el]) == float(torch.tensor(want).to(torch.float16))
        if kk >= 2:
            assert float(w1blk[1, lane, el]) == 0.0                  # the L block multiplies the h half of the patch only
        else:
            check(w1blk[:, lane, el], want)
        pr = rnd.randrange(14)
        if pr == 13 and kk >= 2:                                     # the last tap alone, read as [h | l]: H again, L = 0 against the l half
            want = float(w2[co, 8 * (kk & 1) + el, 2, 8, 0])
            assert float(w2blk[13, 0, lane, el]) == float(torch.tensor(want).to(torch.float16)) and float(w2blk[13, 1, lane, el]) == 0.0
            continue
        tap = pairs[pr][0] if kk < 2 else pairs[pr][1]
        check(w2blk[pr, :, lane, el], float(w2[co, 8 * (kk & 1) + el, tap[0], tap[1], 0]))
    w21blk, _, _, w22blk = e.stage2_tables()[:4]
    w21, w22 = e.stages[2][0], e.stages[3][0]
    assert tuple(w21blk.shape) == (2, 6, 2, 64, 8) and tuple(w22blk.shape) == (2, 24, 2, 64, 8)
    w31blk, w32blk, w41blk = e.conv31_tables()[0], e.conv32t_tables()[0], e.conv41_tables()[0]
    w31, w32, w41 = e.stages[4][0], e.stages[5][0], e.stages[6][0]
    assert tuple(w31blk.shape) == (4, 9, 2, 64, 8) and tuple(w32blk.shape) == (4, 2, 21, 2, 64, 8)
    assert tuple(w41blk.shape) == (8, 9, 2, 2, 64, 8) and w41blk.dtype == torch.float16
    for _ in range(400):
        lane, el = rnd.randrange(64), rnd.randrange(8)
        co, kk = lane & 15, lane >> 4
        nt, pr = rnd.randrange(2), rnd.randrange(6)                  # conv2_1: pair 2 kd + kw / 2, tap b = kw + 1 for kk >= 2
        check(w21blk[nt, pr, :, lane, el], float(w21[16 * nt + co, 8 * (kk & 1) + el, pr // 2, 0, 2 * (pr % 2) + (kk >= 2)]))
        tap = rnd.randrange(24)                                      # conv2_2: one tap per K = 32 block
        check(w22blk[nt, tap, :, lane, el], float(w22[16 * nt + co, 8 * kk + el, tap // 8, tap % 8, 0]))
        nt, tap = rnd.randrange(4), rnd.randrange(9)
        check(w31blk[nt, tap, :, lane, el], float(w31[16 * nt + co, 8 * kk + el, tap // 3, 0, tap % 3]))
        kb, tap = rnd.randrange(2), rnd.randrange(21)
        check(w32blk[nt, kb, tap, :, lane, el], float(w32[16 * nt + co, 32 * kb + 8 * kk + el, tap // 7, tap % 7, 0]))
        nt, tap = rnd.randrange(8), rnd.randrange(9)
        check(w41blk[nt, tap, kb, :, lane, el], float(w41[16 * nt + co, 32 * kb + 8 * kk + el, tap // 3, 0, tap % 3]))


def _regauged(model, alphas):
    """The same function with channel c of layer l carried times alphas[l][c] > 0: BatchNorm's gamma and beta times a (PReLU and the
    pools are positively homogeneous), the next layer's weights on that channel (FC5's columns for the last) divided by it."""
    import copy
    import torch
    from speaker_verification_amd.model import _LAYERS
    m2 = copy.deepcopy(model)
    tags = [t[0] for t in _LAYERS]
    with torch.no_grad():
        for li, tag in enumerate(tags):
            bn, a = getattr(m2, "batch_norm" + tag), alphas[li].to(m2.FC5.weight.device)
            bn.weight.mul_(a)
            bn.bias.mul_(a)
            if li + 1 < len(tags):
                getattr(m2, "conv" + tags[li + 1]).weight.div_(a.view(1, -1, 1, 1, 1))
            else:
                m2.FC5.weight.copy_((m2.FC5.weight.view(128, a.numel(), -1) / a.view(1, -1, 1)).reshape(128, -1))
    return m2


def test_tables_do_not_depend_on_how_a_checkpoint_scales_its_channels():
    """Half pairs have an absolute floor (2^-25) and a ceiling (65 504) where f32 has neither in reach, and a checkpoint is free to
    carry a channel 4 096 times larger with the next layer's weights 4 096 times smaller.  FusedEmbedder fixes each channel's power
    of two from its BatchNorm before it splits the weights: the same network regauged by powers of two (2^-12 .. 2^12 per channel,
    every layer) yields the SAME operand tables bit for bit -- so the same embeddings --, and `act_scale` records the units."""
    import torch
    from speaker_verification_amd.model import FusedEmbedder, _LAYERS, perturb_inference_state, seeded_model
    m = seeded_model(5, 8)
    m.load_state_dict(perturb_inference_state(m.state_dict(), 6))
    m.eval()
    gen = torch.Generator().manual_seed(1)
    alphas = [torch.exp2(torch.randint(-12, 13, (t[2],), generator=gen).float()) for t in _LAYERS]
    m2 = _regauged(m, alphas)
    x = torch.randn(2, 1, 20, 80, 40, generator=gen)
    with torch.no_grad():
        assert torch.equal(m.torch_layers(x), m2.torch_layers(x))          # the same function (powers of two: exactly)
    e1, e2 = FusedEmbedder(m), FusedEmbedder(m2)
    assert all(bool((s == 1).all()) for s in e1.act_scale)                  # an ordinary checkpoint: nothing is rescaled
    for a, s1, s2 in zip(alphas, e1.act_scale, e2.act_scale):
        assert torch.equal(s2 * a, s1)                                      # carried times 1 / a
    for name in ("stage1_tables", "stage2_tables", "conv31_tables", "conv32t_tables", "conv41_tables", "conv42_tables", "fc5_tables"):
        for t1, t2 in zip(getattr(e1, name)(), getattr(e2, name)()):
            assert torch.equal(t1, t2) if torch.is_tensor(t1) else t1 == t2, name
    # without the fix the regauged conv1_2 weights would sit at the halves' floor: up to 2^12 x 2^12 between neighbouring columns
    w = m2.conv1_2.weight.detach()
    assert float(w.abs().amax(dim=(0, 2, 3, 4)).min()) < 2.0 ** -9


def test_checkpoints_outside_the_half_pairs_range_are_refused():
    """A BatchNorm-folded weight that no half can hold (not finite, or 65 504 and more once the channel scales are fixed) raises
    when the operand tables are built -- before any kernel could turn it into an infinity."""
    import pytest
    import torch
    from speaker_verification_amd.model import FusedEmbedder, seeded_model
    m = seeded_model(7, 8).eval()
    with torch.no_grad():
        m.conv2_1.weight[3, 2, 1, 0, 1] = 1.0e6
    with pytest.raises(ValueError, match="65 504"):
        FusedEmbedder(m)
    m = seeded_model(7, 8).eval()
    with torch.no_grad():
        m.batch_norm3_1.running_var[5] = float("nan")
    with pytest.raises(ValueError, match="not finite"):
        FusedEmbedder(m)
    # ... while a channel that is merely carried large is not: its scale is fixed first
    m = seeded_model(7, 8).eval()
    with torch.no_grad():
        m.batch_norm2_1.weight.mul_(1.0e6)
        m.batch_norm2_1.bias.mul_(1.0e6)
        m.conv2_2.weight.div_(1.0e6)
    e = FusedEmbedder(m)
    assert float(e.act_scale[2].max()) == 2.0 ** -20 and float(e.stages[2][0].abs().max()) < 16.0


def test_upload_groups_of_a_host_arena():
    """`VerificationPipeline._upload_groups` (embed_ragged_resident on a host arena): pieces tile the arena, every clip lies
    inside its group's piece, pieces stay under the cap unless one clip alone exceeds it, arena order need not be list order."""
    from speaker_verification_amd.pipeline import VerificationPipeline
    rng = np.random.default_rng(3)
    lens = rng.integers(1000, 90000, size=200).astype(np.int32)
    lens[17] = 400000                                              # longer than the cap: a piece of its own
    slots = (lens + 7) // 8 * 8
    order = rng.permutation(200)                                   # clip k sits at the position of its rank in `order`
    offs = np.zeros(200, dtype=np.int64)
    at = 0
    for k in order:
        offs[k] = at
        at += int(slots[k]) + 8 * int(rng.integers(0, 3))          # gaps between clips are allowed
    cap = 250000
    groups, pieces = VerificationPipeline._upload_groups(offs, lens, at, cap)
    assert pieces[0][0] == 0 and pieces[-1][1] == at and all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
    assert sorted(int(k) for g in groups for k in g) == list(range(200))
    for g, (lo, hi) in zip(groups, pieces):
        assert (offs[g] >= lo).all() and (offs[g] + lens[g] <= hi).all()
        span = int((offs[g] + lens[g]).max() - offs[g].min())
        assert span <= cap or len(g) == 1
    assert any(len(g) == 1 and int(g[0]) == 17 for g in groups)
    assert len(groups) > 10
    # OVERLAPPING clips (50 %-overlapped 3 s windows over one recording): a piece may only end where no clip of it is still
    # running, so that a clip never reaches into a piece that has not been uploaded yet
    n_win, hop, win = 400, 24000, 48000
    offs = np.arange(n_win, dtype=np.int64) * hop
    lens = np.full(n_win, win, dtype=np.int32)
    total = int(offs[-1]) + win
    groups, pieces = VerificationPipeline._upload_groups(offs, lens, total, 500000)
    assert pieces[0][0] == 0 and pieces[-1][1] == total and all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
    assert sorted(int(k) for g in groups for k in g) == list(range(n_win))
    for g, (lo, hi) in zip(groups, pieces):
        assert (offs[g] >= lo).all() and (offs[g] + lens[g] <= hi).all()
    # every window overlaps its successor here: nothing can be cut -- one piece; with gaps every 50 windows: cut only there
    assert len(groups) == 1
    offs2 = offs + (np.arange(n_win) // 50) * win
    groups2, pieces2 = VerificationPipeline._upload_groups(offs2, lens, int(offs2[-1]) + win, 500000)
    assert len(groups2) == 8 and all(len(g) == 50 for g in groups2)
    for g, (lo, hi) in zip(groups2, pieces2):
        assert (offs2[g] >= lo).all() and (offs2[g] + lens[g] <= hi).all()


def test_cube_ring_runs_every_clip_once_in_bounded_memory():
    """`VerificationPipeline._CubeRing` (the ragged paths): batches of any size <= step gather into a ring of 2 x step cubes,
    the network runs over `step` cubes at a time, segments never wrap, and every clip's embedding lands in its own row --
    here with a host stand-in for the two device calls."""
    import types
    from speaker_verification_amd import constants as c
    from speaker_verification_amd.pipeline import VerificationPipeline
    launches = []

    class Eng:
        device = torch.device("cpu")

        @staticmethod
        def cube_gather(feat, idx, frames, out=None, stats=None):
            assert stats is None
            for u in range(feat.shape[0]):
                for k in range(idx.shape[1]):
                    out[u, 0, k] = feat[u, idx[u, k]:idx[u, k] + frames]
            return out

    def embed_features(rows, starts):
        launches.append(rows.shape[0])
        assert torch.equal(starts[0], torch.arange(c.CUBE_CROPS, dtype=torch.int32) * c.CUBE_FRAMES)
        return rows.reshape(rows.shape[0], -1)[:, :128].clone()

    pipe = types.SimpleNamespace(eng=Eng, embed_features=embed_features,
                                 embedder=types.SimpleNamespace(crop_starts=lambda n, dev: (torch.arange(c.CUBE_CROPS, dtype=torch.int32)
                                                                                            * c.CUBE_FRAMES)[None].expand(n, -1).contiguous()))
    rng = np.random.default_rng(5)
    sizes = [3, 4, 1, 4, 4, 2, 4, 3, 4, 4, 1]                            # 34 clips, step 4: the ring (8 cubes) wraps several times
    n = sum(sizes)
    order = torch.from_numpy(rng.permutation(n))
    emb = torch.full((n, 128), float("nan"))
    ring = VerificationPipeline._CubeRing(pipe, 4, emb, order, None)
    assert ring.cubes.shape[0] == 8
    pos, want = 0, torch.empty((n, 128))
    for m in sizes:
        feat = torch.from_numpy(rng.standard_normal((m, 90, c.NUM_COEF)).astype(np.float32))
        idx = torch.from_numpy(rng.integers(0, 10, size=(m, c.CUBE_CROPS)).astype(np.int32))
        for u in range(m):
            want[order[pos + u]] = feat[u, idx[u, 0]:idx[u, 0] + c.CUBE_FRAMES].reshape(-1)[:128]
        ring.push(feat, idx)
        assert ring.at - ring.done < 4                                   # never more than a step pending
        pos += m
    ring.finish()
    assert torch.equal(emb, want)
    assert launches == [4] * 8 + [2] and ring.done == ring.at == n
