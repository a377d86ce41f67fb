"""evaluation.py drop-in (`/root/reference/evaluation.py`): `get_eer_auc`,
`get_and_plot_k_eer_auc`, `Evaluation(...).compute_Similarity(...)`,
`evaluate()`.

The reference scores one (utterance, speaker) pair per sklearn call inside a
Python double loop (evaluation.py:67-84, 112-134); here the whole score matrix
is one MFMA kernel launch (`svk_cosine_scores`), and the embeddings behind it
come from the libsvk network kernels (`model.C3D2.forward` in eval mode on the
device; `dataset_embeddings` for the file-driven entry points).  ROC / EER / AUC
stay on the host exactly as the reference computes them (sklearn + scipy,
evaluation.py:47-52); `get_eer_auc_device` is the device form (`svk_roc_eer`).
"""
import os

import numpy as np
import torch

from .engine import get_engine


def get_eer_auc(label, distance):
    """(eer, auc, fpr, tpr) from flat labels and scores (evaluation.py:47-52)."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn.metrics import roc_auc_score, roc_curve
    fpr, tpr, thresholds = roc_curve(label, distance, pos_label=1)
    auc = roc_auc_score(label, distance)
    eer = brentq(lambda x: 1. - x - interp1d(fpr, tpr)(x), 0., 1.)
    return eer, auc, fpr, tpr


def get_eer_auc_device(label, distance):
    """(eer, auc) like `get_eer_auc`, computed on the GPU (sort + scan + one pass over the ROC
    points): for score sets that should not travel to the host (dev-set scale, 1.8e8 pairs).
    Accepts NumPy arrays or CUDA tensors of any shape; no fpr / tpr arrays are returned."""
    return get_engine().roc_eer(distance, label)


def get_and_plot_k_eer_auc(label, scores, k=1, plot_path='eer_auc.png'):
    """Mean EER / AUC over k consecutive equal slices, printed in percent, ROC
    curves saved to `plot_path` when matplotlib is importable (evaluation.py:11-44).
    Returns (mean_eer, mean_auc) in addition to the reference's prints."""
    step = int(label.shape[0] / float(k))
    eers, aucs, curves = np.zeros((k, 1)), np.zeros((k, 1)), []
    for split_num in range(k):
        lo, hi = split_num * step, (split_num + 1) * step
        eers[split_num], aucs[split_num], fpr, tpr = get_eer_auc(label[lo:hi], scores[lo:hi])
        curves.append((fpr, tpr))
    print("EER=", np.mean(eers) * 100)
    print("AUC=", np.mean(aucs) * 100)
    if plot_path:
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            fig = plt.figure()
            ax = fig.gca()
            for split_num, (fpr, tpr) in enumerate(curves):
                plt.setp(plt.plot(fpr, tpr, label='{} split'.format(split_num)), linewidth=2)
            ax.set_xticks(np.arange(0, 1.1, 0.1))
            ax.set_yticks(np.arange(0, 1.1, 0.1))
            plt.title('ROC with {}-fold cross validation'.format(k))
            plt.xlabel('False Positive Rate')
            plt.ylabel('True Positive Rate')
            plt.grid()
            plt.savefig(plot_path)
            plt.close(fig)
        except ImportError:
            pass
    return float(np.mean(eers)), float(np.mean(aucs))


def score_matrix(test_embeddings, enroll_embeddings):
    """[Nt, D] x [Ns, D] -> [Nt, Ns] float32 cosine scores on the device."""
    return get_engine().cosine_scores(test_embeddings, enroll_embeddings)


class Evaluation:
    """Same constructor and `compute_Similarity` as evaluation.py:55-84.
    `speaker_models_path` may be a directory of `{speaker_id}.pt` tensors (the
    reference's format, Q17) or a dict {speaker_id: (1, D) tensor}."""

    def __init__(self, background_model, speaker_models_path):
        self.model = background_model
        self.speaker_models = {}
        if isinstance(speaker_models_path, dict):
            for key, value in speaker_models_path.items():
                self.speaker_models[key] = torch.as_tensor(value)
        else:
            for file in sorted(os.listdir(speaker_models_path)):          # explicit order
                if file.endswith('.pt'):
                    self.speaker_models[file.replace('.pt', '')] = torch.load(
                        os.path.join(speaker_models_path, file), map_location="cpu", weights_only=True)
        self._enroll = None

    def _enroll_matrix(self):
        if self._enroll is None:
            eng = get_engine()
            rows = [m.detach().reshape(1, -1).to(torch.float32) for m in self.speaker_models.values()]
            self._enroll = eng.to_device(torch.cat(rows, dim=0))
        return self._enroll

    def embed(self, utterance):
        """`self.model(utterance, development=False)` in eval mode on the device (evaluation.py:68-69): for a `model.C3D2`
        that is the seven libsvk network kernels (the cube read as feature rows, model.C3D2.forward)."""
        eng = get_engine()
        self.model.eval()
        self.model.to(eng.device)
        with torch.no_grad():
            return self.model(eng.to_device(utterance, torch.float32), development=False)

    def compute_Similarity(self, utterance, type='cosine_similarity'):
        """(similarity_vec, assigned_speaker_vec), both float64 of length n_speakers."""
        speaker_features = self.embed(utterance)
        if type == 'cosine_similarity':
            scores = get_engine().cosine_scores(speaker_features[:1], self._enroll_matrix())
            similarity_vec = scores[0].to("cpu").numpy().astype(np.float64)
            assigned_speaker_vec = np.zeros(len(self.speaker_models))
            assigned_speaker_vec[np.argmax(similarity_vec)] = 1
            return similarity_vec, assigned_speaker_vec

    def score_all(self, cubes, batch=256):
        """[N, 1, 20, 80, 40] cubes -> [N, n_speakers] float32 scores (device), batched."""
        outs = []
        for lo in range(0, len(cubes), batch):
            outs.append(get_engine().cosine_scores(self.embed(cubes[lo:lo + batch]), self._enroll_matrix()))
        return torch.cat(outs, dim=0)


def labels_from_ids(test_ids, speaker_ids):
    """One-hot truth rows like evaluation.py:130-132."""
    speaker_ids = list(speaker_ids)
    labels = np.zeros((len(test_ids), len(speaker_ids)))
    for i, sid in enumerate(test_ids):
        labels[i, speaker_ids.index(sid)] = 1
    return labels


def _read_batch(dataset, lo, hi):
    """Files [lo, hi) of an `AudioDataset` as ONE arena + 16-byte-aligned clip offsets + lengths.  16 kHz mono 16-bit files
    (what the reference's tree holds, vad.py:10-22) are read straight into an int16 arena -- no float copy on the host, the
    `/ 32768` of `librosa.load` (utils.py:170-173) is folded into the front end's filterbank weights; a batch with any other
    rate / channel count falls back to `load_signal`'s float32 (resampled on the device)."""
    import wave
    from . import constants as c
    paths = [os.path.join(dataset.audio_dir, dataset.sound_files[i]) for i in range(lo, hi)]
    heads = []
    for path in paths:
        with wave.open(path, "rb") as wf:
            heads.append((wf.getnchannels(), wf.getsampwidth(), wf.getframerate(), wf.getnframes()))
    plain = all(h[:3] == (1, 2, c.SAMPLE_RATE) for h in heads)
    if plain:
        lens = np.array([h[3] for h in heads], dtype=np.int32)
        align = 8
    else:
        sigs = [np.asarray(dataset.load_signal(i), dtype=np.float32) for i in range(lo, hi)]
        lens = np.array([x.size for x in sigs], dtype=np.int32)
        align = 4
    slots = (lens.astype(np.int64) + align - 1) // align * align
    offs = np.concatenate([[0], np.cumsum(slots)[:-1]]).astype(np.int64)
    buf = np.zeros(int(slots.sum()), dtype=np.int16 if plain else np.float32)
    for k, path in enumerate(paths):
        if plain:
            with wave.open(path, "rb") as wf:
                buf[offs[k]:offs[k] + lens[k]] = np.frombuffer(wf.readframes(int(lens[k])), dtype=np.int16)
        else:
            buf[offs[k]:offs[k] + lens[k]] = sigs[k]
    return buf, offs, lens, plain


def dataset_embeddings(dataset, model, batch=256):
    """Embeddings [len(dataset), 128] (device) of every file of a `load_data.AudioDataset`, in order:
    the per-item chain of the reference (load_data.py:50-87 `load_wav` -> `lmfe`; utils.py:382-397 CMVN;
    utils.py:351-379 FeatureCube with crop starts from the GLOBAL NumPy RNG, drawn in file order;
    `model(cube, development=False)`) run `batch` files at a time: one ragged front-end launch, one CMVN and the seven
    libsvk network kernels per batch -- the cube is never built (`svk_c3d2_stage1` reads feature rows + crop starts)."""
    from . import _lib
    from . import constants as c
    from .engine import spec_from_seconds
    eng = get_engine()
    model = model.to(eng.device).eval()
    n = len(dataset)
    out = torch.empty((n, 128), dtype=torch.float32, device=eng.device)
    if c.DERIVATIVE:
        # three-channel cubes (utils.py:385-391): the per-item transform chain, the torch module's forward batched
        for lo in range(0, n, batch):
            cubes = np.stack([np.asarray(dataset[i][0], dtype=np.float32) for i in range(lo, min(n, lo + batch))])
            with torch.no_grad():
                out[lo:lo + len(cubes)] = model(eng.to_device(cubes), development=False)
        return out
    specs = {plain: spec_from_seconds(c.SAMPLE_RATE, c.FRAME_LEN, c.FRAME_STEP, c.NUM_FFT, c.NUM_COEF, c.NUM_COEF, _lib.OUT_LMFE,
                                      input_scale=1.0 / 32768.0 if plain else 1.0) for plain in (True, False)}
    embed = model.fused_inference()
    for lo in range(0, n, batch):
        hi = min(n, lo + batch)
        buf, offs, lens, plain = _read_batch(dataset, lo, hi)
        spec = specs[plain]
        frames = [spec.num_frames(int(v)) for v in lens]
        feat, n_frames, _ = eng.features(buf, spec, lengths=lens, offsets=offs, max_frames=max(frames))
        if c.NORMALIZE:
            eng.cmvn_(feat, n_frames, variance=True)
        # utils.py:372, one draw per file in file order (numpy raises for clips of <= 80 frames, as there)
        idx = np.stack([np.random.randint(T - c.CUBE_FRAMES, size=c.CUBE_CROPS) for T in frames]).astype(np.int32)
        out[lo:hi] = embed.embed_features(feat, idx)
    return out


def load_indexed_labels(path):
    """{speaker id: class index} for `create_dataset`.  The reference keeps it as a PICKLED dict inside
    `50_first_ids.npy` (`np.load(..., allow_pickle=True).item()`, evaluation.py:100, model.py:360); unpickling runs
    whatever the file says, so this build never does it.  Sources, in order: a `.json` file of the same stem; else the
    id list of the same stem (`50_first_ids.txt`, which carries the same information): speaker id = the first 7
    characters of each path (load_data.py:73), class index = rank among the sorted ids.  The labels only ride along
    in the dataset items; enrolment and evaluation key on the ids themselves."""
    import json
    stem = os.path.splitext(path)[0]
    if os.path.exists(stem + '.json'):
        with open(stem + '.json') as fh:
            return json.load(fh)
    if os.path.exists(stem + '.txt'):
        ids = sorted({str(line)[0:7] for line in np.atleast_1d(np.genfromtxt(stem + '.txt', dtype='str'))})
        return {sid: k for k, sid in enumerate(ids)}
    raise FileNotFoundError(f"{stem}.json / {stem}.txt not found ({path} is a pickle: not loaded)")


def _evaluate_files(k, plot_path):
    """evaluation.py:90-146 as written: checkpoint, id list, id table, WAV tree and enrolled models
    under `constants.ROOT` / `constants.DATA_ORIGIN`."""
    from . import constants as c
    from .model import C3D2
    from .utils import create_dataset
    model_path = os.path.join(c.ROOT, 'Models/model_14_percent_best_so_far.pt')
    checkpoint = torch.load(model_path, map_location="cpu", weights_only=True)
    model = C3D2(100, 1).load_checkpoint(checkpoint)
    dir_path = os.path.join(c.ROOT, 'speaker_models')
    test_set = os.path.join(c.ROOT, '50_first_ids.txt')
    indexed_labels = load_indexed_labels(c.ROOT + '/50_first_ids.npy')
    dataset = create_dataset(indexed_labels=indexed_labels, origin_file_path=test_set)
    ev = Evaluation(model, dir_path)
    speaker_model_ids = list(ev.speaker_models.keys())
    emb = dataset_embeddings(dataset, model)
    scores = get_engine().cosine_scores(emb, ev._enroll_matrix()).to("cpu").numpy().astype(np.float64)
    labels = np.zeros_like(scores)
    correct = 0
    ids = np.array(speaker_model_ids)
    for i in range(len(dataset)):
        current_id = dataset.sound_files[i][0:7]
        closest = speaker_model_ids[int(np.argmax(scores[i]))]
        print('correct speaker {} , the speaker was closer to {}'.format(current_id, closest))
        correct += int(current_id == closest)
        labels[i][np.where(current_id == ids)] = 1                   # an id that was never enrolled: all zeros
    eer, auc = get_and_plot_k_eer_auc(labels.flatten(), scores.flatten(), k=k, plot_path=plot_path)
    accuracy = correct * 100 / len(dataset)
    print(f'Accuracy: {accuracy}%')
    return {"eer": eer, "auc": auc, "accuracy": accuracy, "scores": scores, "labels": labels,
            "speaker_ids": speaker_model_ids, "test_ids": [f[0:7] for f in dataset.sound_files]}


def evaluate(model=None, cubes=None, test_ids=None, speaker_models=None, k=1, plot_path='eer_auc.png'):
    """`evaluate()` -- no arguments, like evaluation.py:90-146: read the checkpoint, the id list, the WAVs and
    the enrolled `{id}.pt` models from the paths in `constants`, score every utterance against every
    enrolled speaker (one batched front end + network + ONE cosine launch instead of the reference's
    per-utterance, per-speaker loop), print the reference's lines, save the ROC plot.
    `evaluate(model, cubes, test_ids, speaker_models)` is the same loop on in-memory data.
    Returns a dict (the reference returns None): eer, auc, accuracy, scores, labels."""
    if model is None and cubes is None:
        return _evaluate_files(k, plot_path)
    ev = Evaluation(model, speaker_models)
    speaker_ids = list(ev.speaker_models.keys())
    scores = ev.score_all(cubes).to("cpu").numpy().astype(np.float64)
    labels = labels_from_ids(test_ids, speaker_ids)
    correct = int(sum(speaker_ids[int(np.argmax(scores[i]))] == test_ids[i] for i in range(len(test_ids))))
    eer, auc = get_and_plot_k_eer_auc(labels.flatten(), scores.flatten(), k=k, plot_path=plot_path)
    accuracy = correct * 100 / max(1, len(test_ids))
    print(f'Accuracy: {accuracy}%')
    return {"eer": eer, "auc": auc, "accuracy": accuracy, "scores": scores, "labels": labels}
