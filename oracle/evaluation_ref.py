"""CPU oracle for the reference's FILE-DRIVEN entry points -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, one utterance at a time like the reference (paths relative to `/root/reference/`):
  * `model.create_speaker_models()`  (`model.py:351-388`): for every WAV of the enrolment list
    `dataset[i]` -> `model(cube, development=False)` -> `torch.save` to `{id}.pt`, the LAST utterance of a
    speaker winning (Q17);
  * `evaluation.evaluate()`  (`evaluation.py:90-146`): `dataset[i]` -> `compute_Similarity` against every
    enrolled model (in `os.listdir` order) -> one-hot labels -> EER / AUC / accuracy;
  * the dataset item they share (`load_data.py:50-87` + `utils.py:18-31,351-397`): `load_wav` (librosa:
    a 16 kHz mono 16-bit file comes back as int16 / 32768 in float32), `lmfe(signal, 16000, 0.025, 0.01,
    40, 1024)`, CMVN only when `NORMALIZE`, `FeatureCube((80, 40, 20))` with crop starts from the GLOBAL
    NumPy RNG (`utils.py:372`), in file order.

Parity status: PINNED by `tests/golden/round2.npz` (`eval_*`), which `tools/make_golden.py` produced by
running the reference's own `create_speaker_models()` and `evaluate()` on the tree
`synth.write_verification_tree` writes (librosa / torchvision are absent: the generator supplied the WAV
reader and `Compose` described in its docstrings).
"""
import os
import wave

import numpy as np

from . import model_ref, scoring_ref, speechpy_ref


def load_wav(path):
    with wave.open(path, "rb") as wf:
        assert wf.getframerate() == 16000 and wf.getnchannels() == 1 and wf.getsampwidth() == 2
        pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)
    return pcm.astype(np.float32) / np.float32(32768.0)


def dataset_item(path, normalize=False):
    """(1, 20, 80, 40) float32 cube of one WAV; consumes one `np.random.randint` draw (utils.py:372)."""
    feat = speechpy_ref.lmfe(load_wav(path), 16000, 0.025, 0.01, 40, 1024)       # load_data.py:64-70
    if normalize:
        feat = speechpy_ref.cmvn(feat, variance_normalization=True)              # utils.py:394-395
    idx = np.random.randint(feat.shape[0] - 80, size=20)
    return model_ref.feature_cube(feat, idx)


def create_speaker_models(data_dir, rel_paths, state, normalize=False):
    """{speaker id: (1, 128) float32} -- model.py:374-388."""
    store = {}
    for rel in rel_paths:
        cube = dataset_item(os.path.join(data_dir, rel), normalize)
        store[rel[0:7]] = model_ref.c3d2_embed(state, cube[None]).numpy()
    return store


def evaluate(data_dir, rel_paths, state, speaker_models, speaker_order, normalize=False):
    """(scores [n, n_spk] float64, labels, accuracy in percent, eer, auc) -- evaluation.py:107-146."""
    enroll = [speaker_models[s] for s in speaker_order]
    ids = np.array(speaker_order)
    scores, labels, correct = [], [], 0
    for rel in rel_paths:
        cube = dataset_item(os.path.join(data_dir, rel), normalize)
        emb = model_ref.c3d2_embed(state, cube[None]).numpy()
        sims, _ = scoring_ref.compute_similarity(emb, enroll)
        scores.append(sims)
        current = rel[0:7]
        correct += int(current == speaker_order[int(np.argmax(sims))])
        lab = np.zeros_like(sims)
        lab[np.where(current == ids)] = 1
        labels.append(lab)
    scores, labels = np.array(scores), np.array(labels)
    eer, auc = scoring_ref.k_fold_eer_auc(labels.flatten(), scores.flatten(), k=1)
    return scores, labels, correct * 100 / len(rel_paths), eer, auc
